// decode_wide.hip -- CTC prefix beam search for beam widths above the wave-per-sequence kernels' 256 (decode.hip; 64 until round 6).
//
// radian/decode.py:145 slices `sort_labelings()[:beam_width]` with whatever --beam-width the user gave (basecall.py:32), so
// a width of 64 or 100 or 500 is a valid run of the reference.  decode.hip keeps a sequence's whole beam set in the lanes and LDS of
// one to ten waves, which ends at 256 beams (a beam set is built by the lanes of one wave, in at most four parts); this kernel is the general form: one workgroup of four waves per sequence, the
// kept beams and the 5 W candidates of a time step in an HBM scratch block per sequence (L2-resident: <= 300 W bytes), only
// the ranking keys in LDS (40 W bytes).  Same semantics, phase for phase, as beam_search_kernel:
//   A  candidate q = 5 i + k (k = 0: the copy of kept beam i, decode.py:150-175; k = 1..4: its extension by label k - 1,
//      :186-201) in the reference's dict insertion order; the LM gate (decode.py:79-96) on the label the candidate consumes
//   B/C an extension whose labeling is already kept merges with that beam's copy (logaddexp of pr_non_blank and pr_total,
//      decode.py:172-175,199-201); the merged entry sits where the first of the two was inserted
//   D/E rank by (pr_total descending, insertion order ascending) = Python's stable sort (decode.py:35-39); the best W become
//      the new beam set, extensions get their canonical trie id (the parent's child id, or a fresh one)
// Labeling identity is the same HBM trie as decode.hip's (childtab / backptr); "is this labeling kept right now" is a per-node
// slot map in HBM instead of an LDS id table.  Scores: decode_common.h -- glibc arithmetic gives the reference's bits.
// Not here: hashed long contexts (a mode with no reference behaviour; rd_decode_dev refuses the combination).
#include "decode_common.h"

namespace {

constexpr int kWideTPB = 256;

struct __attribute__((aligned(16))) WBeam {
    double ptot, pb, pnb;   // pr_total, pr_blank, pr_non_blank (log)   decode.py:20-25
    int last, len;          // last label (-1: empty labeling), labeling length
    int node;               // canonical trie id
    unsigned hist;          // last 16 labels, 2 bits each (LM context)
};
static_assert(sizeof(WBeam) == 48, "WBeam is three 16-B words");

struct WideArgs {
    char* scratch;          // per sequence: rd_wide_scratch_bytes(W) (common.h)
    size_t stride;
    int* slot_of_node;      // per trie node (indexed like backptr): slot of the kept beam that carries it, -1 = not kept now
};

template <bool GX>
__device__ __forceinline__ double lae_m(double x, double y, const uint64_t* gx_exp)
{
    if constexpr (GX) return lae_gx(x, y, gx_exp);
    else return lae(x, y);
}

template <typename PT, bool LM, bool GX>
__global__ __launch_bounds__(kWideTPB) void beam_search_wide_kernel(DecodeArgs a, WideArgs w)
{
    extern __shared__ __attribute__((aligned(16))) double skey[];   // [5 W]: a survivor's pr_total at its candidate index, NaN elsewhere
    __shared__ double lp[64][5];
    __shared__ double praw[LM ? 64 : 1][5];
    __shared__ double sent[LM ? 64 : 1];
    __shared__ uint64_t gx_lds[GX ? 256 : 1];
    __shared__ int s_nvalid, s_next_id, s_missed;

    const int tid = threadIdx.x;
    const int seq = blockIdx.x;
    const int T = a.seq_len[seq];
    const int W = a.W;
    const PT* __restrict__ probs = (const PT*)a.probs;
    const int64_t row_a = a.seq_off[seq];
    const int64_t row_b = a.seq_off2 ? a.seq_off2[seq] : row_a;
    const int split = a.seq_off2 ? a.seq_split[seq] : 0;
    int4* childtab = a.childtab + a.node_off[seq];
    int* backptr = a.backptr + a.node_off[seq];
    int* slot_of = w.slot_of_node + a.node_off[seq];
    const unsigned ctx_mask = LM ? ((a.k >= 16) ? 0xffffffffu : ((1u << (2 * a.k)) - 1u)) : 0u;

    char* base = w.scratch + (size_t)seq * w.stride;
    WBeam* const bm = (WBeam*)base;                          // [2][W]
    double* const c_ptot = (double*)(bm + 2 * (size_t)W);    // [5 W] each
    double* const c_pb = c_ptot + 5 * (size_t)W;
    double* const c_pnb = c_pb + 5 * (size_t)W;
    int* const c_pj = (int*)(c_pnb + 5 * (size_t)W);         // extension: slot of the kept beam with the same labeling, or -1
    int* const c_src = c_pj + 5 * (size_t)W;                 // entry's labeling: >= 0 that kept beam's; -1 parent + label; -2 entry merged away
    int* const mq = c_src + 5 * (size_t)W;                   // [W]: the extension that merges into kept beam j's copy, or -1

    if constexpr (GX)
        for (int i = tid; i < 256; i += kWideTPB) gx_lds[i] = g_gm_exp_tab[i];
    const uint64_t* const gx_exp = gx_lds;

    // decode.py:128-132: the empty labeling with pr_blank = pr_total = log(1)
    if (tid == 0) {
        WBeam b;
        b.ptot = 0.0;
        b.pb = 0.0;
        b.pnb = -INFINITY;
        b.last = -1;
        b.len = 0;
        b.node = 0;
        b.hist = 0u;
        bm[0] = b;
        childtab[0] = make_int4(0, 0, 0, 0);
        backptr[0] = 0;
        slot_of[0] = 0;
        s_next_id = 1;
        s_missed = 0;
        s_nvalid = 0;
    }
    int nb = 1;     // beams currently kept (workgroup-uniform)
    int cur = 0;
    __syncthreads();

    for (int t0 = 0; t0 < T; t0 += 64) {
        // ---- per-tile prepass: one thread per time step: the 5 log-probabilities (decode.py:165,168,193,195) and, with an LM, the
        //      entropy of the renormalised base distribution (decode.py:135-138, numpy-1.19 dtypes as in decode.hip)
        if (tid < 64) {
            const int t = t0 + tid;
            if (t < T) {
                const PT* __restrict__ prow = probs + ((t < split ? row_a : row_b) + t) * 5;
                double s4 = 0.0;
#pragma unroll 1
                for (int c = 0; c < 5; c++) {
                    const double pc = (double)prow[c];
                    lp[tid][c] = safe_log<GX>(pc);
                    if constexpr (LM) {
                        praw[tid][c] = pc;
                        if (c < 4) s4 = c == 0 ? pc : s4 + pc;
                    }
                }
                if constexpr (LM) {
                    const double s = s4;
                    double ent = 0.0;
                    bool any = false;
#pragma unroll 1
                    for (int c = 0; c < 4; c++) {
                        const double pc = praw[tid][c];
                        double n;
                        if (s == 0.0) n = pc;
                        else if constexpr (sizeof(PT) == 4) n = (double)((float)pc / (float)s);
                        else n = pc / s;
                        if (n > 0) {
                            double v = n * log_m<GX>(n);
                            ent = any ? ent + v : v;
                            any = true;
                        }
                    }
                    sent[tid] = any ? -ent : 0.0;
                }
            }
        }
        __syncthreads();

        const int tend = (T - t0) < 64 ? (T - t0) : 64;
        for (int tt = 0; tt < tend; tt++) {
            const WBeam* os = bm + (size_t)cur * W;
            WBeam* ns = bm + (size_t)(cur ^ 1) * W;
            const int ncand = 5 * nb;
            const double lp_blank = lp[tt][4];
            bool s_open = false;
            if constexpr (LM) s_open = sent[tt] > a.s_thr;

            // ---------------- A: candidate scores; which extension equals which kept labeling?
            for (int q = tid; q < ncand; q += kWideTPB) {
                const int i = q / 5, k = q - 5 * i;
                const WBeam b = os[i];
                const bool is_copy = k == 0;
                const int c = is_copy ? b.last : k - 1;          // label whose probability this candidate consumes
                double lpc = c < 0 ? -INFINITY : lp[tt][c < 0 ? 0 : c];
                if constexpr (LM) {
                    // decode.py:157-163 (copy: context excludes the last label) and :180-184 (extend)
                    const int need = is_copy ? a.k + 1 : a.k;
                    if (s_open && c >= 0 && b.len >= need) {
                        const unsigned ctx = (is_copy ? (b.hist >> 2) : b.hist) & ctx_mask;
                        const bool gate = (a.lm_gate[ctx >> 5] >> (ctx & 31)) & 1u;
                        if (gate) {
                            // combine_dists decode.py:52-64
                            const double r = a.lm_table[(size_t)ctx * 4 + c];
                            double val;
                            if constexpr (sizeof(PT) == 4) {
                                const float f0 = (float)praw[tt][0], f1 = (float)praw[tt][1], f2 = (float)praw[tt][2], f3 = (float)praw[tt][3];
                                const float bp = ((f0 + f1) + f2) + f3;
                                const float sb = (float)praw[tt][c] / bp;
                                val = ((r + (double)sb) / 2.0) * (double)bp;
                            } else {
                                const double bp = ((praw[tt][0] + praw[tt][1]) + praw[tt][2]) + praw[tt][3];
                                const double sb = praw[tt][c] / bp;
                                val = ((r + sb) / 2.0) * bp;
                            }
                            lpc = safe_log<GX>(val);
                        }
                    }
                }
                if (is_copy) {
                    const double pnb_c = (b.last >= 0) ? b.pnb + lpc : -INFINITY;
                    const double pb_c = b.ptot + lp_blank;
                    c_ptot[q] = lae_m<GX>(pb_c, pnb_c, gx_exp);     // decode.py:174-175
                    c_pb[q] = pb_c;
                    c_pnb[q] = pnb_c;
                    c_pj[q] = -1;
                    c_src[q] = i;
                    mq[i] = -1;
                } else {
                    const double v = ((b.last == k - 1) ? b.pb : b.ptot) + lpc;   // decode.py:192-195
                    c_ptot[q] = v;
                    c_pb[q] = -INFINITY;
                    c_pnb[q] = v;
                    const int x = ((const int*)&childtab[b.node])[k - 1];
                    c_pj[q] = x != 0 ? slot_of[x] : -1;
                    c_src[q] = -1;
                }
            }
            __syncthreads();
            // ---------------- B: at most one extension has kept beam j's labeling (its parent and label are unique)
            for (int q = tid; q < ncand; q += kWideTPB) {
                const int pj = c_pj[q];
                if (pj >= 0) mq[pj] = q;
            }
            __syncthreads();
            // ---------------- C: merge (decode.py:172-175 and :198-201 hit the same dict entry); the entry lives where it was inserted first
            for (int j = tid; j < nb; j += kWideTPB) {
                const int qe = mq[j];
                if (qe >= 0) {
                    const int q = 5 * j;
                    const double v = c_ptot[qe];
                    const double P = lae_m<GX>(c_ptot[q], v, gx_exp);
                    const double Q = lae_m<GX>(c_pnb[q], v, gx_exp);
                    const double cb = c_pb[q];
                    const int first = q < qe ? q : qe, other = q < qe ? qe : q;
                    c_ptot[first] = P;
                    c_pnb[first] = Q;
                    c_pb[first] = cb;
                    c_src[first] = j;
                    c_src[other] = -2;
                }
            }
            __syncthreads();
            // ---------------- D: the ranking's keys.  Only entries that can reach the top W: with W beams kept, every kept labeling
            //                  survives as an entry >= tau (its copy's pr_total >= pr_total_old + log p(blank)), see decode.hip Phase D
            const double tau = (nb == W) ? os[nb - 1].ptot + lp_blank : -INFINITY;
            int nv = 0;
            for (int q = tid; q < ncand; q += kWideTPB) {
                const bool valid = c_src[q] != -2;
                const double key = c_ptot[q];
                nv += valid ? 1 : 0;
                skey[q] = (valid && key >= tau) ? key : __builtin_nan("");
            }
            if (nv) atomicAdd(&s_nvalid, nv);
            for (int j = tid; j < nb; j += kWideTPB) slot_of[os[j].node] = -1;     // (the beams that stay re-enter below)
            __syncthreads();
            const int nvalid = s_nvalid;
            const int nb_new = nvalid < W ? nvalid : W;
            // ---------------- E: rank (pr_total descending, insertion order ascending: decode.py:35-39) and build the new beam set
            for (int q = tid; q < ncand; q += kWideTPB) {
                const double key = skey[q];
                if (!(key == key)) continue;
                int rank = 0;
                for (int p = 0; p < ncand; p++) {
                    const double kv = skey[p];
                    rank += ((kv > key) || (kv == key && p < q)) ? 1 : 0;
                }
                if (rank >= W) continue;
                WBeam nbm;
                nbm.ptot = c_ptot[q];
                nbm.pb = c_pb[q];
                nbm.pnb = c_pnb[q];
                const int src = c_src[q];
                if (src >= 0) {
                    const WBeam b = os[src];
                    nbm.last = b.last;
                    nbm.len = b.len;
                    nbm.node = b.node;
                    nbm.hist = b.hist;
                } else {
                    const int i = q / 5, cl = q - 5 * i - 1;
                    const WBeam par = os[i];
                    int id = ((const int*)&childtab[par.node])[cl];
                    if (id == 0) {      // never created: a fresh canonical id
                        id = atomicAdd(&s_next_id, 1);
                        backptr[id] = (par.node << 2) | cl;
                        ((int*)&childtab[par.node])[cl] = id;
                        childtab[id] = make_int4(0, 0, 0, 0);
                    }
                    nbm.last = cl;
                    nbm.len = par.len + 1;
                    nbm.node = id;
                    nbm.hist = (par.hist << 2) | (unsigned)cl;
                    if constexpr (LM) {
                        // sparse model: the reference looks model[context] up for every kept labeling of >= k labels at every later step
                        if (a.lm_missing && nbm.len >= a.k && t0 + tt + 1 < T) {
                            const unsigned cx = nbm.hist & ctx_mask;
                            if ((a.lm_missing[cx >> 5] >> (cx & 31)) & 1u) s_missed = 1;
                        }
                    }
                }
                ns[rank] = nbm;
                slot_of[nbm.node] = rank;
            }
            nb = nb_new;
            cur ^= 1;
            __syncthreads();
            if (tid == 0) s_nvalid = 0;      // (read by everyone before the barrier above; written again after the next step's first one)
        }
    }

    // ---------------- traceback of the best labeling (slot 0 = rank 0; decode.py:207-210)
    __syncthreads();
    if (tid == 0) {
        const WBeam fs = bm[(size_t)cur * W];
        int n = fs.node;
        uint8_t* out = a.labels + a.label_off[seq];
        for (int p = fs.len - 1; p >= 0; p--) {
            const int bp = backptr[n];
            out[p] = (uint8_t)(bp & 3);
            n = bp >> 2;
        }
        a.label_len[seq] = (LM && s_missed) ? -1 : fs.len;
        if (a.best_score) a.best_score[seq] = fs.ptot;
    }
}

template <typename PT>
int launch_wide_pt(hipStream_t st, const DecodeArgs& a, const WideArgs& w, int n_seq, bool lm)
{
    const size_t lds = (size_t)5 * a.W * sizeof(double);
    const dim3 grid((unsigned)n_seq), block(kWideTPB);
    if (lm) {
        if (a.glibc_math) hipLaunchKernelGGL((beam_search_wide_kernel<PT, true, true>), grid, block, lds, st, a, w);
        else hipLaunchKernelGGL((beam_search_wide_kernel<PT, true, false>), grid, block, lds, st, a, w);
    } else {
        if (a.glibc_math) hipLaunchKernelGGL((beam_search_wide_kernel<PT, false, true>), grid, block, lds, st, a, w);
        else hipLaunchKernelGGL((beam_search_wide_kernel<PT, false, false>), grid, block, lds, st, a, w);
    }
    RD_HIP(hipGetLastError());
    return RD_OK;
}

}  // namespace

// Widths above rd_decode_lane_width() (64).  `a` is the argument block rd_decode_dev has filled (decode.hip); the scratch block and the
// slot map come out of the context's workspaces, which -- like the trie -- one beam search uses at a time.
int rd_decode_wide_launch(rd_ctx* ctx, hipStream_t st, const void* args, int ptype, int n_seq, int64_t total_nodes, bool lm)
{
    const DecodeArgs& a = *(const DecodeArgs*)args;
    RD_REQUIRE(a.W <= RD_WIDE_MAX_W, "beam_width %d out of range [1,%d]", a.W, RD_WIDE_MAX_W);
    RD_REQUIRE(!(lm && a.hashed), "beam widths above %d do not combine with hashed long contexts (rd_load_lm_hashed)", 64);
    WideArgs w;
    w.stride = rd_wide_scratch_bytes(a.W);
    if (ctx->ws_wide.reserve(w.stride * (size_t)n_seq)) return RD_ERR_NOMEM;
    if (ctx->ws_wide_slot.reserve((size_t)total_nodes * sizeof(int))) return RD_ERR_NOMEM;
    w.scratch = (char*)ctx->ws_wide.p;
    w.slot_of_node = ctx->ws_wide_slot.as<int>();
    return ptype == 1 ? launch_wide_pt<double>(st, a, w, n_seq, lm) : ptype == 2 ? launch_wide_pt<_Float16>(st, a, w, n_seq, lm)
                                                                                 : launch_wide_pt<float>(st, a, w, n_seq, lm);
}
