"""Per-map-size PMC table of the Winograd convolution: every stride-1 3x3 shape of BASELINE configs[1] that
the Winograd kernels serve, B = 8, REPS launches each, alone on the device.

  python tools/wino_pmc_table.py run                   (under rocprofv3 --kernel-trace --pmc ...; tools/_pmc_wino_table.sh)
  python tools/wino_pmc_table.py parse <dir> > json    (dir holds sq/ fetch/ write/ passes)
"""
import collections, csv, glob, json, os, sys

# (H, W, Cin, Cout, residual, pool, launches of this shape per registration iteration, where)
SHAPES = [
    (352, 1216, 64, 64, True, 1, 2, "MiniResNet block 1 (full resolution)"),
    (176, 608, 64, 64, True, 1, 3, "MiniResNet blocks 2-3"),
    (88, 304, 64, 64, True, 1, 13, "MiniResNet tail / fuse / head convolutions"),
    (88, 304, 128, 128, False, 2, 11, "agent conv 1 (10 steps) + decoder"),
    (88, 304, 64, 128, True, 1, 10, "agent conv 0, projected half (+ cached image half as residual)"),
    (88, 304, 128, 64, False, 1, 1, "decoder 128 -> 64"),
    (44, 152, 128, 128, False, 1, 10, "agent conv 2"),
    (44, 152, 128, 128, False, 2, 10, "agent conv 3 (+ pool)"),
    (22, 76, 128, 128, False, 1, 10, "agent conv 4"),
    (22, 76, 128, 128, False, 2, 10, "agent conv 5 (+ pool)"),
    (11, 38, 128, 128, False, 1, 20, "agent convs 6-7"),
]
REPS = 4


def run():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from cmr_agent_amd import ops
    from cmr_agent_amd.models._pack import winograd_u
    dev, B = "cuda:0", 8
    acc = 0.0
    for H, W, ci, co, res, pool, _, _ in SHAPES:
        x = torch.randn(B, H, W, ci, device=dev)
        u = winograd_u(torch.randn(co, ci, 3, 3, device=dev) * 0.05)
        b = torch.randn(co, device=dev)
        r = torch.randn(B, H, W, co, device=dev) if res else None
        for _ in range(REPS):
            y = ops.conv3x3_wino(x, u, b, co, 0.2, res=r, pool=pool)
        torch.cuda.synchronize()
        acc += float(y.sum())
    print(acc)


def parse(root):
    def disp(sub):
        f = sorted(glob.glob(root + "/" + sub + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)[-1]
        per = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            if "conv3x3_wino" not in r["Kernel_Name"]:
                continue
            d = per.setdefault(int(r["Dispatch_Id"]), {"kernel": r["Kernel_Name"].split("(")[0].replace("(anonymous namespace)::", "").replace("void ", "")})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            d["_ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        rows = [per[k] for k in sorted(per)]
        assert len(rows) == REPS * len(SHAPES), (sub, len(rows))
        return rows
    sq, fe, wr = disp("sq"), disp("fetch"), disp("write")
    out = {"_note": "rocprofv3 --kernel-trace --pmc passes (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU | FETCH_SIZE | "
                    "WRITE_SIZE) of tools/wino_pmc_table.py run: B = 8, each shape alone on the device, means over the last %d of %d launches; "
                    "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs); FETCH_SIZE in KiB doubled (gfx950), WRITE_SIZE in "
                    "KiB; issued TFLOP/s = 16/36 of the algorithmic FLOPs / duration under the profiler" % (REPS - 1, REPS), "rows": []}
    tot_busy = tot_w = 0.0
    for i, (H, W, ci, co, res, pool, per_iter, where) in enumerate(SHAPES):
        sl = slice(i * REPS + 1, (i + 1) * REPS)
        mean = lambda rows, key: sum(r.get(key, 0.0) for r in rows[sl]) / (REPS - 1)
        ns, cyc = mean(sq, "_ns"), mean(sq, "GRBM_GUI_ACTIVE") / 8.0
        busy = mean(sq, "SQ_VALU_MFMA_BUSY_CYCLES") / (cyc * 1024)
        fl = 2.0 * 9 * ci * co * 8 * H * W
        alg_bytes = 4.0 * 8 * H * W * (ci + co / (pool * pool) + (co if res else 0)) + 4 * 9 * ci * co
        row = dict(shape="%dx%d %d->%d%s%s" % (H, W, ci, co, " +res" if res else "", " +pool" if pool == 2 else ""), where=where,
                   kernel=sq[sl][0]["kernel"], launches_per_iteration=per_iter, us=round(ns / 1e3, 1), mfma_busy=round(busy, 4),
                   clock_ghz=round(cyc / ns, 3), issued_tflops=round(fl * 16 / 36 / ns / 1e3, 1), algorithmic_gflop=round(fl / 1e9, 2),
                   hbm_fetch_mb=round(2 * 1024 * mean(fe, "FETCH_SIZE") / 1e6, 1), hbm_write_mb=round(1024 * mean(wr, "WRITE_SIZE") / 1e6, 1),
                   algorithmic_mb=round(alg_bytes / 1e6, 1), valu_insts_per_mfma_cycle=None)
        out["rows"].append(row)
        tot_busy += busy * ns * per_iter
        tot_w += ns * per_iter
    out["mfma_busy_time_weighted_over_an_iteration"] = round(tot_busy / tot_w, 4)
    out["winograd_us_per_iteration_alone"] = round(tot_w / 1e3, 1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else parse(sys.argv[2])
