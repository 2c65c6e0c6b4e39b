"""Parse the HIP runtime's own graph dump (DEBUG_HIP_GRAPH_DOT_PRINT=1 writes graph_<pid>_dot_print_<n> into the working directory at
instantiation): node index, the runtime's stream assignment, whether the node carries a completion signal, kernel; cross-stream edges.
python tools/graph_dot_parse.py <dump file>"""
import re
import sys

txt = open(sys.argv[1]).read()
nodes = {}
for m in re.finditer(r'"(graph_\d+_node_(\d+))"\[[^\]]*?label="([^"]*)"\]', txt, re.S):
    lab = m.group(3).split("\n")
    name = lab[1] if len(lab) > 1 else ""
    name = re.sub(r"^_ZN\d*_?GLOBAL__N_1\d+", "", name)
    name = re.sub(r"^_ZN2at6native\d+", "at::", name)
    sid = re.search(r"StreamId:(\d+)", m.group(3))
    sig = re.search(r"SignalIsRequired: (\w+)", m.group(3))
    extra = lab[2] if name == "MEMCPY" and len(lab) > 2 else ""
    nodes[m.group(1)] = (int(m.group(2)), int(sid.group(1)) if sid else -1, sig.group(1) == "true" if sig else None, (name + " " + extra)[:70])
edges = [(a, b) for a, b in re.findall(r'"(graph_\d+_node_\d+)"\s*->\s*"(graph_\d+_node_\d+)"', txt)]
print("%d nodes, %d edges; per stream: %s" % (len(nodes), len(edges), {s: sum(1 for n in nodes.values() if n[1] == s) for s in sorted(set(n[1] for n in nodes.values()))}))
print("cross-stream edges (from -> to):")
for a, b in edges:
    na, nb = nodes[a], nodes[b]
    if na[1] != nb[1]:
        print("  %4d (s%d, signal %s) %-40s -> %4d (s%d) %s" % (na[0], na[1], na[2], na[3][:40], nb[0], nb[1], nb[3][:40]))
print("nodes in the dump's order:")
for k, (i, s, sig, name) in sorted(nodes.items(), key=lambda kv: kv[1][0]):
    print("  %4d  s%d  %s  %s" % (i, s, "SIGNAL" if sig else "      ", name))
