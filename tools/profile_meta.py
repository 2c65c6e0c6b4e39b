"""Freshness sidecar of a committed profile: `python tools/profile_meta.py <profile file> <source file> ...` writes `<profile>.meta.json`
= {"sources": {repo-relative path: sha256}, "made": date}.  bench.py reports a figure read from a committed profile (roofline.frac of
the headline = the dominant kernel inside the timed hipGraph replays; roofline.traffic = PMC bytes) ONLY while every listed source still
has the recorded hash -- the GPU box has no .git, so the check is by content, not by commit."""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sha(path):
    return hashlib.sha256(open(os.path.join(ROOT, path), "rb").read()).hexdigest()


if __name__ == "__main__":
    prof, srcs = sys.argv[1], sys.argv[2:]
    meta = {"sources": {os.path.relpath(os.path.abspath(s), ROOT): sha(os.path.relpath(os.path.abspath(s), ROOT)) for s in srcs},
            "made": time.strftime("%Y-%m-%d %H:%M:%S")}
    json.dump(meta, open(prof + ".meta.json", "w"), indent=1)
    print(prof + ".meta.json", len(srcs), "sources")
