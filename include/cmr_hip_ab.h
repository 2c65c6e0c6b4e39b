/* A/B switches of libcmr_hip_ab.so -- NOT part of the product ABI.
 *
 * cmr_agent_amd/lib/libcmr_hip_ab.so is the same source tree compiled with -DCMR_AB_SWITCHES: it exports everything include/cmr_hip.h
 * declares PLUS the process-global kernel-variant switches below, which tests use to compare two kernels bit for bit and tools/*_bench.py
 * to time them.  The product (cmr_agent_amd/*, bench.py, Train_*.py, Test_Agent.py) loads libcmr_hip.so, which has none of these
 * symbols and no mutable state: its dispatch thresholds are compile-time constants (tests/test_abi.py asserts both).  Every switch
 * returns the previous setting unless stated otherwise. */
#ifndef CMR_HIP_AB_H
#define CMR_HIP_AB_H
#include "cmr_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Debug / benchmark switch: 0 routes the contiguous [rows][64] -> [rows][64 | 32] calls of cmr_linear_f32 through the generic
 * weight-stationary kernel instead of the row-streaming fast path (bit-identical results); returns the previous setting. */
int cmr_set_linear_row64(int on);

/* The same kind of switch for the register-weights kernel (K = 64, n_out <= 64, one source, at least min_rows rows; min_rows <= 0 keeps the
 * threshold): bit-identical to the weight-stationary kernel.  Returns the previous on / off setting. */
int cmr_set_linear_wreg(int on, int64_t min_rows);

/* Process-wide switch between the two Winograd kernels for maps of >= 200 tiles (1 = wave-specialised persistent kernel,
 * the default; 0 = 4-wave workgroups for every map): A/B measurements and tests only.  Returns the previous setting. */
int cmr_set_wino_variant(int wave_specialised);

/* MFMA waves per SIMD of the wave-specialised Winograd kernel: 1 = 8-wave workgroups (4 multiply, 4 help: the default), 2 = 12-wave
 * workgroups (8 multiply -- position row x cout tile -- 4 help; round-5 experiment, 10 - 25 % SLOWER: profiles/r05_wino_mw_ab.txt).  Same
 * products in the same order: bit-identical results.  A/B measurements and tests only.  Returns the previous setting. */
int cmr_set_wino_mfma_waves(int per_simd);

/* Process-wide switch of the bf16 convolution's kernel choice for 128-cout layers without residual / table operand (stride 1,
 * Cin = 64 | 128): matrix_class = 1 (default) routes maps of at least min_tiles 8x32-pixel tiles (x Cout / 128; min_tiles <= 0 keeps the
 * current threshold) to the register-tiled kernel that streams the weight fragments from L2 (conv3x3_bf16_mm_kernel), 0 keeps the
 * two-team kernel everywhere.  Same products, accumulated per 64-channel chunk: results agree to fp32 rounding of the sums.  A/B
 * measurements and tests only.  Returns CMR_OK. */
int cmr_set_conv_bf16_variant(int matrix_class, int min_tiles);

/* Process-wide switch between the two softmax-attention kernels (1 = v_mfma_f32_16x16x4_f32 for Q K^T and P V, the default;
 * 0 = one query per 4 lanes on the vector ALUs): A/B measurements and tests only.  Returns the previous setting. */
int cmr_set_mha_variant(int mfma);

/* A/B switch: 1 (default) = the LDS-staged kernel for Cin 64 / 128 on maps of >= 4096 pixels (a ring of input rows in LDS, every
 * tap an LDS address), 0 = the direct kernel everywhere (operands by dword loads).  Same sums in a different order (results agree
 * to fp32 rounding); returns the previous setting. */
int cmr_set_wgrad_variant(int lds_staged);

/* A/B switch of the bf16 3x3 weight gradient on Cin 128 / Cout % 64 == 0 maps of >= 32 768 pixels: 1 (default, the product library's only
 * choice) = second generation (rows staged through registers as they lie in memory, operands through ds_read_b64_tr_b16, 64 couts per
 * 8-wave workgroup), 2 = third generation (rows by LDS-DMA into a raw ring, one input row per iteration; on the same strips bit-identical
 * to 1; round-5 experiment, measured slower: profiles/r05_wgrad_dma_ablate.txt), 16 + mask = its timing-only ablations, 0 = the
 * first-generation kernel everywhere (rows transposed by the lanes on the way into LDS; same products, other summation order).
 * Returns the previous setting. */
int cmr_set_wgrad_bf16_variant(int generation);

/* bf16 weight gradient: strips per workgroup the row count of a strip is sized for -- per_workgroup > 0: third generation (default 4),
 * < 0: second generation (- per_workgroup; default 8; 4 measured 4 % faster alone and no different in the update: profiles/r05_wgrad_dma_ablate.txt),
 * 0 keeps both.  Other strips = other partial sums: results agree to fp32 rounding.  Returns the third generation's previous setting. */
int cmr_set_wgrad_bf16_strips(int per_workgroup);

/* A/B switch: 1 (default) = row maps of >= 65 536 rows with n, k multiples of 32 up to 128 take the LDS-staged kernel (whole-row
 * float4 staging, the full gradient per workgroup), 2 = from 8192 rows on (tests), 0 = the direct kernel everywhere.  Same sums in
 * another order; returns the previous on / off setting. */
int cmr_set_linear_wgrad_variant(int lds_staged);

#ifdef __cplusplus
}
#endif
#endif
