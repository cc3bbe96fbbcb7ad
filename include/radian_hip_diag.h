/*
 * radian_hip_diag.h -- measurement and diagnostic entry points of libradian_hip.so.
 *
 * NOT part of the drop-in boundary (include/radian_hip.h): nothing here has a counterpart in comprna/radian and nothing here
 * changes a result -- launch shapes for A/B runs, HIP-event kernel timers, read-outs of the reads pipeline's self-measured policy.
 * Callers: bench.py (roofline), tools/, tests/.  The product's command line (radian_amd/basecall.py) calls none of them
 * (tests/test_abi_cpu.py checks that).  Same conventions as radian_hip.h: 0 or a negative RD_ERR_*, rd_last_error().
 */
#ifndef RADIAN_HIP_DIAG_H
#define RADIAN_HIP_DIAG_H

#include "radian_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- launch shapes (A/B measurements; bit-identical results) -------------------------------- */
/* Workgroup shape of the fp32 matrix-product kernels (no effect on results; for measurements): 0 (default) = 128 time steps x 256
 * channels per 256-thread workgroup, two workgroups per CU; 1 = 256 x 256 per 512-thread workgroup, one per CU (the weight
 * tile is shared by twice the rows: a third less LDS-DMA volume per FLOP, no second workgroup to run under an epilogue).
 * Applies to the exact-fp32 mode only: the split-f16 kernels exist in shape 0, the bf16x3 kernels in shape 1 (rd_set_precision). */
int rd_set_conv_shape(rd_ctx* ctx, int shape);
/* Block 0's first conv (one input channel: three multiply-adds and a ReLU per output; model.py:71, keras-tcn conv1D_0 of
 * residual_block_0) computed inside the kernel of the block's second conv instead of by a kernel of its own (no effect on
 * results: bit-identical; exact-fp32 mode, dilation <= 2): 1 (default) / 0.  For measurements and the identity test. */
int rd_set_conv_fuse(rd_ctx* ctx, int on);
/* Launch shape of the beam search (no effect on results; no reference counterpart): 0 = chosen per launch (default).  Widths
 * above 12: several waves per sequence while the launch leaves SIMDs idle, else two candidates per lane; 1 / 2 pin either.
 * Widths up to 6 (the reference's default, basecall.py:32) run two sequences per wave (one candidate per lane of a half-wave);
 * widths 7..12 run ONE sequence per wave under form 0 -- two per wave (two candidates per lane) exists and measured slower, so
 * only form 3 selects it; 3 = two per wave whenever the width allows (up to 12), 4 = always one per wave.  Widths above
 * rd_decode_lane_width() run on the general kernel whatever the form.  5 = every launch (widths up to 256, no hashed
 * contexts) through the work-queue kernel with 16 waves' worth of workgroups: the form the reads pipeline uses, with the partition's
 * resident count, for a group that holds more sequences than its decode partition.  For tests and measurements. */
int rd_set_decode_form(rd_ctx* ctx, int form);

/* Workspace one beam-search launch may ask for, in bytes (default 24 GiB; 0 restores it).  A launch needs (1 + W * rows) trie nodes of
 * 20-24 B per sequence for all its sequences at once; launches that would need more are cut into runs of sequences that go one after the
 * other on the same stream and share the workspace (at least one sequence per run).  No effect on results: the test that cuts small
 * launches into many runs sets this. */
int rd_set_trie_budget(rd_ctx* ctx, int64_t bytes);

/* ---- numerics diagnostic -------------------------------------------------------------------- */
/* Diagnostic of rd_set_precision mode 2 (bf16x3): split n fp32 values on the device exactly as the kernels do; terms_out[t * n + i] is the bf16 bit
 * pattern of term t (0 hi, 1 mid, 2 lo) of values[i]. */
int rd_split3(rd_ctx* ctx, const float* values, size_t n, uint16_t* terms_out);

/* ---- reads pipeline: policy and counters ---------------------------------------------------- */
/* The global-mode groups close by COVERAGE: when the forward rows gathered so far take the next group's forwards as long as the
 * beam search of this group's longest read will take (a read's search is one serial chain; radian/basecall.py:99-109 runs it
 * inline, here it runs under the next reads' forwards).  Both sides of that rule are measured by the context itself with HIP
 * events -- ns per forward row (per matrix-product mode) and us per time step of a group's longest chain (per beam width,
 * arithmetic, LM, and per occupancy of the decode partition: on_partition = 1..3 waves per SIMD, 0 = the whole chip, where the
 * built-in figure stays in force, scaled by the measured forward pace) -- starting from built-in figures for the exact-fp32
 * mode.  Read-out for tools and tests: 0 = not measured yet; *rows_per_step = the rule in force (forward rows per time step
 * of the longest read). */
int rd_pipe_policy_read(rd_ctx* ctx, int beam_width, int on_partition, int use_lm, double* ns_per_row, double* us_per_step,
                        int64_t* rows_per_step);
/* Counters of the reads-level pipeline over the context's life (no reference counterpart; for tools and tests): out[0..n) of
 * { batches submitted, batches delivered, groups whose beam search was launched, of those: global-mode groups that held more sequences than
 * their decode partition keeps resident and were searched through the work queue, groups closed at that limit instead }. */
int rd_pipe_stats(rd_ctx* ctx, int64_t* out, int n);

/* ---- kernel timing on the launch stream (HIP events) --------------------------------------- */
#define RD_TIMER_CONV 0   /* dilated conv 256->256 (MFMA), the dominant kernel */
#define RD_TIMER_DECODE 1 /* beam search */
#define RD_TIMER_HEAD 2   /* dense head + softmax */
#define RD_TIMER_IN 3     /* block-0 first conv (C_in = 1) */
int rd_timer_enable(rd_ctx* ctx, int which, int max_launches); /* 0 disables */
int rd_timer_read(rd_ctx* ctx, int which, double* total_ms, int* launches, double* flops, double* bytes);
/* The recorded launches one by one (up to cap): duration, algorithmic FLOPs, and for RD_TIMER_CONV the epilogue variant -- 0 relu (a
 * block's first conv), 1 res_ident (second conv, identity residual), 2 res_match (block 0's second conv: 1x1 match conv, and with
 * rd_set_conv_fuse 1 the block's first conv inside) -- so that a roofline fraction can be stated per variant and launch-weighted. */
int rd_timer_read_launches(rd_ctx* ctx, int which, int cap, float* ms_out, double* flops_out, int32_t* tag_out, int* n_out);

#ifdef __cplusplus
}
#endif
#endif /* RADIAN_HIP_DIAG_H */
