"""gpurun_out/pmc_train/<tag>/{fetch,write}/**/counter_collection.csv -> HBM bytes per C-ABI CALL of the fused-layer / convolution entry
points of a training step (JSON on stdout): the kernels of an entry point are summed and divided by the dispatch count of its main kernel.
FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled on gfx950 (128-B requests tallied at 64 B, MI355X_MICROARCH.md, HBM section)."""
import csv, glob, json, collections, os, sys
root = sys.argv[1]
ENTRIES = {   # entry point -> (main kernel, other kernels of the same call)
    "cmr_bn_linear_bwd_f32": ("bn_linear_bwd_kernel", ("blb_reduce_kernel", "blb_coef_final_kernel", "blb_seg_reduce_kernel")),
    "cmr_linear_bn_fwd_f32": ("bn_linear_fwd_kernel", ("bn_stats_merge_kernel",)),
    "cmr_conv3x3_wino_nhwc_f32": ("conv3x3_wino_ws_kernel", ("conv3x3_wino_kernel",)),
    "cmr_conv3x3_wino_stats_nhwc_f32": ("conv3x3_wino_ws_kernel", ("conv3x3_wino_kernel",)),      # (the same kernel: the launches with the BatchNorm sums are not told apart)
    "cmr_conv3x3_wino_bnbwd_nhwc_f32": ("conv3x3_wino_ws_kernel", ("conv3x3_wino_kernel",)),
    "cmr_conv3x3_wgrad_f32": ("conv3x3_wgrad_reduce_kernel", ("conv3x3_wgrad_kernel", "conv3x3_wgrad_lds_kernel", "conv3x3_wgrad_s2_kernel")),      # (one reduction per call)
    "cmr_affine_act_f32": ("affine_act_kernel", ()),
    # round 5: the bf16 agent update's convolution / weight-gradient entry points and the BatchNorm sweeps
    "cmr_conv3x3_wgrad_bias_bf16_f32": ("conv3x3_wgrad_reduce_kernel", ("conv3x3_wgrad_bf16_tr_kernel", "conv3x3_wgrad_bf16_kernel", "conv_bias_reduce_kernel")),
    "cmr_conv3x3_bf16_nhwc_f32": ("conv3x3_bf16_", ()),
    "cmr_bn_bwd_f32": ("bn_bwd_apply_kernel", ("bn_bwd_partial_kernel", "bn_bwd_final_kernel")),
    "cmr_bn_stats_f32": ("bn_stats_final_kernel", ("bn_stats_partial_kernel",)),
}


def collect(sub, counter):
    acc = collections.defaultdict(lambda: [set(), 0.0])
    files = sorted(glob.glob(root + "/" + sub + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    for f in files[-1:]:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            acc[r["Kernel_Name"]][0].add(r["Dispatch_Id"])
            acc[r["Kernel_Name"]][1] += float(r["Counter_Value"])
    return acc


fe, wr = collect("fetch", "FETCH_SIZE"), collect("write", "WRITE_SIZE")
out = {"_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate eager runs of the bench command (side streams off); counters are KiB; "
                "FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM section); per C-ABI call = all kernels of "
                "the entry point / dispatches of its main kernel"}
for entry, (main, others) in ENTRIES.items():
    def tot(acc, names):
        n, v = 0, 0.0
        for k, (ids, val) in acc.items():
            if any(nm in k for nm in names):
                v += val
                if main in k:
                    n += len(ids)
        return n, v
    nf, f = tot(fe, (main,) + others)
    nw, w = tot(wr, (main,) + others)
    # the two 3x3 weight-gradient entry points share their reduction kernel: an entry point counts only when one of its OWN kernels ran
    own = [o for o in others if "reduce" not in o and "final" not in o]
    if "wgrad" in entry and own and not any(any(o in k for o in own) for k in fe):
        continue
    if nf and nw:
        out[entry] = {"calls_profiled": nf, "fetch_bytes_per_call": 2 * 1024 * f / nf, "write_bytes_per_call": 1024 * w / nw,
                      "hbm_bytes_per_call": 2 * 1024 * f / nf + 1024 * w / nw}
print(json.dumps(out, indent=1))
