// decode.hip -- CTC prefix beam search with the k-mer RNA-LM gate, one wavefront per sequence.
//
// Replaces radian/decode.py:100-212 (beam_search) and :42-96 (LM gate) of the reference.
//
// Mapping to gfx950: a sequence is a T-long serial dependency chain, so each sequence gets ONE
// 64-lane wave (workgroup = 1 wave; many waves per CU hide each other's latency).  Within a time
// step the 5*W candidate entries (for each kept beam: its copy + 4 extensions, decode.py:150-201)
// are spread over lanes: candidate q = 5*i + k lives in slot q/64 of lane q%64 (R slots per lane:
// R=1 for W<=12, R=2 for W<=25, R=4 for W<=51).  Candidate order q is exactly the reference's dict
// insertion order, which is what Python's stable sort falls back to on ties (decode.py:38).
//
// Exact labeling identity (the dict keyed by tuples, decode.py:171-201) is kept with a per-sequence
// trie whose node ids are canonical: every kept beam carries the ids of its four children
// (0 = never created); a new id is allocated only when a never-created child enters the top W, and
// the trie is written through to HBM so that a labeling that leaves the beam and later re-enters is
// found again under the same id.  "copy of beam j" and "extension of beam i by c" collide iff
// child[c][i] == node[j]; the two probabilities are then combined with logaddexp exactly as the
// reference does.  Back-pointers (parent<<2 | label) in HBM give the final labeling by traceback.
//
// Scores are float64 log-probabilities; log / log1p / exp are ROCm's double-precision device
// functions (<= 1 ulp from glibc), so scores agree with the reference to a few ulp and the emitted
// labeling is identical unless two beams tie within that distance.
#include "common.h"

#include <math.h>

namespace {

constexpr double kLogE2 = 0.693147180559945309417232121458176568;

// numpy npy_logaddexp (decode.py:172-201 call np.logaddexp on python floats)
__device__ __forceinline__ double lae(double x, double y)
{
    if (x == y) return x + kLogE2;
    double hi = fmax(x, y), lo = fmin(x, y);
    return hi + log1p(exp(lo - hi));
}

// The workgroup IS one wave (launch bounds 64), so LDS hand-offs between lanes need no s_barrier and -- the point -- no
// s_waitcnt vmcnt(0): __syncthreads() would also wait for the acknowledgement of the trie stores to HBM of every time
// step (~1-2 us each on a loaded chip).  LDS operations of one wave execute in order; the fences keep the compiler from
// moving LDS accesses across the hand-off.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// decode.py:16-17
__device__ __forceinline__ double safe_log(double x) { return x == 0.0 ? -INFINITY : log(x); }

template <int WM>
struct BeamState {
    double ptot[WM], pb[WM], pnb[WM];
    int node[WM], len[WM], last[WM];
    unsigned hist[WM];
    int child[4][WM];
};

template <int R>
struct Cfg {
    static constexpr int NC = 64 * R;       // candidate capacity
    static constexpr int WM = (64 * R) / 5; // max beam width
};

struct DecodeArgs {
    const void* probs;
    const int64_t* seq_off;
    const int64_t* seq_off2;   // nullable: rows t >= seq_split[i] come from row seq_off2[i] + t (streamed forward)
    const int32_t* seq_split;
    const int32_t* seq_len;
    const int64_t* node_off;
    const int64_t* label_off;
    int W;
    // LM
    const double* lm_table;
    const uint32_t* lm_gate;
    int k;
    double s_thr;
    // trie in HBM
    int4* childtab;
    int* backptr;
    // out
    uint8_t* labels;
    int32_t* label_len;
    double* best_score;
};

template <typename PT, int R, bool LM>
__global__ __launch_bounds__(64) void beam_search_kernel(DecodeArgs a)
{
    constexpr int WM = Cfg<R>::WM;
    const int lane = threadIdx.x;
    const int seq = blockIdx.x;
    const int T = a.seq_len[seq];
    const PT* __restrict__ probs = (const PT*)a.probs;
    const int64_t row_a = a.seq_off[seq];
    const int64_t row_b = a.seq_off2 ? a.seq_off2[seq] : row_a;
    const int split = a.seq_off2 ? a.seq_split[seq] : 0;
    int4* __restrict__ childtab = a.childtab + a.node_off[seq];
    int* __restrict__ backptr = a.backptr + a.node_off[seq];
    const int W = a.W;
    const unsigned ctx_mask = LM ? ((a.k >= 16) ? 0xffffffffu : ((1u << (2 * a.k)) - 1u)) : 0u;

    __shared__ BeamState<WM> st[2];
    __shared__ double cpy_pnb[WM], cpy_tot[WM], cpy_pb[WM], mb_v[WM], mP[WM], mQ[WM];
    __shared__ int mb_q[WM];
    __shared__ int d_copy[WM], d_par[WM], d_c[WM], rk_owner[WM];
    __shared__ double lp[64][5];
    __shared__ double praw[LM ? 64 : 1][5];
    __shared__ double sent[LM ? 64 : 1];

    // decode.py:128-132: the empty labeling with pr_blank = pr_total = log(1)
    if (lane == 0) {
        st[0].ptot[0] = 0.0;
        st[0].pb[0] = 0.0;
        st[0].pnb[0] = -INFINITY;
        st[0].node[0] = 0;
        st[0].len[0] = 0;
        st[0].last[0] = -1;
        st[0].hist[0] = 0u;
        for (int c = 0; c < 4; c++) st[0].child[c][0] = 0;
        childtab[0] = make_int4(0, 0, 0, 0);
        backptr[0] = 0;
    }
    int nb = 1;        // beams currently kept (wave-uniform)
    int next_id = 1;   // next free trie node id (wave-uniform)
    int cur = 0;
    wave_sync();

    for (int t0 = 0; t0 < T; t0 += 64) {
        // ---- per-tile prepass: one lane per time step computes the 5 log-probabilities (decode.py:165,168,193,195
        //      take math.log of mat[t][c]) and, with an LM, the entropy of the renormalised base distribution
        //      (decode.py:135-138).
        {
            const int t = t0 + lane;
            if (t < T) {
                double p[5];
#pragma unroll
                for (int c = 0; c < 5; c++) p[c] = (double)probs[((t < split ? row_a : row_b) + t) * 5 + c];
#pragma unroll
                for (int c = 0; c < 5; c++) lp[lane][c] = safe_log(p[c]);
                if constexpr (LM) {
#pragma unroll
                    for (int c = 0; c < 5; c++) praw[lane][c] = p[c];
                    // normalise(): sum(dist) in float64 (numpy-1.19 semantics); float32 rows divide in float32
                    double s = ((p[0] + p[1]) + p[2]) + p[3];
                    double ent = 0.0;
                    bool any = false;
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        double n;
                        if (s == 0.0) n = p[c];
                        else if constexpr (sizeof(PT) == 4) n = (double)((float)p[c] / (float)s);
                        else n = p[c] / s;
                        if (n > 0) {
                            double v = n * log(n);
                            ent = any ? ent + v : v;
                            any = true;
                        }
                    }
                    sent[lane] = any ? -ent : 0.0;
                }
            }
        }
        wave_sync();

        const int tend = (T - t0) < 64 ? (T - t0) : 64;
        for (int tt = 0; tt < tend; tt++) {
            BeamState<WM>& os = st[cur];
            BeamState<WM>& ns = st[cur ^ 1];
            const int ncand = 5 * nb;

            // ---------------- Phase A: candidate scores -------------------------------------------------
            bool valid[R];
            int bi[R], kk[R], pj[R], dcopy[R];
            double c_ptot[R], c_pnb[R], c_pb[R];
            const double lp_blank = lp[tt][4];
#pragma unroll
            for (int s = 0; s < R; s++) {
                const int q = s * 64 + lane;
                valid[s] = q < ncand;
                const int i = valid[s] ? q / 5 : 0;
                const int k = q - 5 * (q / 5);
                bi[s] = i;
                kk[s] = k;
                pj[s] = -1;
                dcopy[s] = -1;
                const double ptot_i = os.ptot[i], pb_i = os.pb[i], pnb_i = os.pnb[i];
                const int last_i = os.last[i];
                const int len_i = os.len[i];
                const int c = (k == 0) ? last_i : k - 1;  // label whose probability this candidate consumes
                double lpc = (c >= 0) ? lp[tt][c] : -INFINITY;
                if constexpr (LM) {
                    // decode.py:157-163 (copy: context excludes the last label) and :180-184 (extend)
                    const int need = (k == 0) ? a.k + 1 : a.k;
                    if (valid[s] && c >= 0 && len_i >= need) {
                        const unsigned h = os.hist[i];
                        const unsigned ctx = ((k == 0) ? (h >> 2) : h) & ctx_mask;
                        const bool gate = ((a.lm_gate[ctx >> 5] >> (ctx & 31)) & 1u) && (sent[tt] > a.s_thr);
                        if (gate) {
                            // combine_dists decode.py:52-64
                            const double r = a.lm_table[(size_t)ctx * 4 + c];
                            double val;
                            if constexpr (sizeof(PT) == 4) {
                                const float f0 = (float)praw[tt][0], f1 = (float)praw[tt][1], f2 = (float)praw[tt][2],
                                            f3 = (float)praw[tt][3];
                                const float bp = ((f0 + f1) + f2) + f3;
                                const float sb = (float)praw[tt][c] / bp;
                                val = ((r + (double)sb) / 2.0) * (double)bp;
                            } else {
                                const double bp = ((praw[tt][0] + praw[tt][1]) + praw[tt][2]) + praw[tt][3];
                                const double sb = praw[tt][c] / bp;
                                val = ((r + sb) / 2.0) * bp;
                            }
                            lpc = safe_log(val);
                        }
                    }
                }
                if (k == 0) {
                    // COPY decode.py:150-175
                    const double pnb_c = (last_i >= 0) ? pnb_i + lpc : -INFINITY;
                    const double pb_c = ptot_i + lp_blank;
                    c_pnb[s] = pnb_c;
                    c_pb[s] = pb_c;
                    c_ptot[s] = 0.0;  // lae(pb_c, pnb_c) below
                    dcopy[s] = i;
                } else {
                    // EXTEND decode.py:186-201
                    const double v = ((last_i == k - 1) ? pb_i : ptot_i) + lpc;
                    c_pnb[s] = v;
                    c_pb[s] = -INFINITY;
                    c_ptot[s] = v;
                }
            }
            // total of the copy candidates: logaddexp(pr_blank, pr_non_blank) decode.py:174-175
#pragma unroll
            for (int s = 0; s < R; s++) {
                const double tot = lae(c_pb[s], c_pnb[s]);
                if (kk[s] == 0) {
                    c_ptot[s] = tot;
                    if (valid[s]) {
                        cpy_pnb[bi[s]] = c_pnb[s];
                        cpy_tot[bi[s]] = tot;
                        cpy_pb[bi[s]] = c_pb[s];
                        mb_q[bi[s]] = -1;
                    }
                }
            }
            wave_sync();

            // ---------------- Phase C: which extension equals which kept labeling? ----------------------
            const int node_reg = lane < nb ? os.node[lane] : -1;   // beam j's trie id lives in lane j
#pragma unroll
            for (int s = 0; s < R; s++) {
                const int x = (valid[s] && kk[s] > 0) ? os.child[kk[s] - 1][bi[s]] : 0;
                int found = -1;
#pragma unroll
                for (int j = 0; j < WM; j++)   // lanes >= nb hold -1: constant lane selects, no loop-carried scalar
                    if (__builtin_amdgcn_readlane(node_reg, j) == x) found = j;
                if (x != 0 && found >= 0) {
                    pj[s] = found;
                    mb_q[found] = s * 64 + lane;
                    mb_v[found] = c_ptot[s];
                }
            }
            wave_sync();
            bool any_merge = false;
#pragma unroll
            for (int s = 0; s < R; s++) {
                const bool m = valid[s] && ((kk[s] == 0) ? (mb_q[bi[s]] >= 0) : (pj[s] >= 0));
                any_merge |= __any(m);
            }
            if (any_merge) {
                // merged entry: pr_non_blank = lae(copy.pnb, v); pr_total = lae(copy.total, v)   (decode.py:172-175,199-201)
#pragma unroll
                for (int s = 0; s < R; s++) {
                    double x = -INFINITY, y = -INFINITY;
                    const bool is_copy = kk[s] == 0;
                    const bool m = valid[s] && (is_copy ? (mb_q[bi[s]] >= 0) : (pj[s] >= 0));
                    if (m) {
                        if (is_copy) { x = c_ptot[s]; y = mb_v[bi[s]]; }
                        else { x = cpy_pnb[pj[s]]; y = c_ptot[s]; }
                    }
                    const double r = lae(x, y);
                    if (m) {
                        if (is_copy) mP[bi[s]] = r;
                        else mQ[pj[s]] = r;
                    }
                }
                wave_sync();
#pragma unroll
                for (int s = 0; s < R; s++) {
                    const int q = s * 64 + lane;
                    if (valid[s]) {
                        if (kk[s] == 0) {
                            const int qe = mb_q[bi[s]];
                            if (qe >= 0) {
                                if (q < qe) { c_ptot[s] = mP[bi[s]]; c_pnb[s] = mQ[bi[s]]; }
                                else valid[s] = false;
                            }
                        } else if (pj[s] >= 0) {
                            const int qc = 5 * pj[s];
                            if (q < qc) {
                                c_ptot[s] = mP[pj[s]];
                                c_pnb[s] = mQ[pj[s]];
                                c_pb[s] = cpy_pb[pj[s]];
                                dcopy[s] = pj[s];
                            } else valid[s] = false;
                        }
                    }
                }
            }

            // ---------------- Phase D: rank by (pr_total desc, insertion order asc)  decode.py:35-39,145 ------
            // rank = number of candidates ahead.  Candidate qq's key is broadcast out of its lane's register with
            // v_readlane (a scalar operand of the compares): no LDS round trip per candidate.
            int nvalid = 0;
            double key[R];
#pragma unroll
            for (int s = 0; s < R; s++) {
                key[s] = valid[s] ? c_ptot[s] : __builtin_nan("");
                nvalid += __popcll(__ballot(valid[s]));
            }
            // Fast count: candidates with a strictly greater key, four broadcasts per trip (lanes past ncand hold NaN and
            // count for nothing).  Two candidates get the same count iff their keys are equal, so a collision among the
            // counts below W -- found by letting the lanes claim rk_owner[count] -- means a tie that matters; only then is
            // the count redone with the insertion-order rule (exact 0 probabilities make such ties; softmax rows do not).
            int rank[R];
#pragma unroll
            for (int s = 0; s < R; s++) rank[s] = 0;
#pragma unroll
            for (int s2 = 0; s2 < R; s2++) {
                const int cnt = (ncand - s2 * 64) < 64 ? (ncand - s2 * 64) : 64;   // wave-uniform
                const int klo = __double2loint(key[s2]), khi = __double2hiint(key[s2]);
                for (int l = 0; l < cnt; l += 4) {
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int ll = (l + u) & 63;
                        const double kv = __hiloint2double(__builtin_amdgcn_readlane(khi, ll), __builtin_amdgcn_readlane(klo, ll));
#pragma unroll
                        for (int s = 0; s < R; s++) rank[s] += (kv > key[s]) ? 1 : 0;
                    }
                }
            }
            bool tie = false;
#pragma unroll
            for (int s = 0; s < R; s++)
                if (valid[s] && rank[s] < W) rk_owner[rank[s]] = s * 64 + lane;
            wave_sync();
#pragma unroll
            for (int s = 0; s < R; s++) tie |= valid[s] && rank[s] < W && rk_owner[rank[s]] != s * 64 + lane;
            if (__any(tie)) {
#pragma unroll
                for (int s = 0; s < R; s++) rank[s] = 0;
#pragma unroll
                for (int s2 = 0; s2 < R; s2++) {
                    const int cnt = (ncand - s2 * 64) < 64 ? (ncand - s2 * 64) : 64;
                    const int klo = __double2loint(key[s2]), khi = __double2hiint(key[s2]);
                    for (int l = 0; l < cnt; l++) {
                        const double kv = __hiloint2double(__builtin_amdgcn_readlane(khi, l), __builtin_amdgcn_readlane(klo, l));
                        const int qq = s2 * 64 + l;
#pragma unroll
                        for (int s = 0; s < R; s++) {
                            const int q = s * 64 + lane;
                            const bool ahead = (kv > key[s]) || (kv == key[s] && qq < q);
                            rank[s] += ahead ? 1 : 0;
                        }
                    }
                }
            }
            const int nb_new = nvalid < W ? nvalid : W;

            // ---------------- Phase E: scatter the kept candidates to their new beam slot ---------------------
#pragma unroll
            for (int s = 0; s < R; s++) {
                if (valid[s] && rank[s] < W) {
                    const int r = rank[s];
                    ns.ptot[r] = c_ptot[s];
                    ns.pb[r] = c_pb[s];
                    ns.pnb[r] = c_pnb[s];
                    d_copy[r] = dcopy[s];
                    d_par[r] = bi[s];
                    d_c[r] = kk[s] - 1;
                }
            }
            wave_sync();

            // ---------------- Phase F: trie ids for the new beam set ------------------------------------------
            int my_node = 0, my_par = 0, my_c = 0;
            bool fresh = false, reload = false;
            const bool is_new_ext = (lane < nb_new) && (d_copy[lane] < 0);
            if (is_new_ext) {
                my_par = d_par[lane];
                my_c = d_c[lane];
                my_node = os.child[my_c][my_par];
                fresh = (my_node == 0);
                reload = !fresh;
            }
            const unsigned long long fmask = __ballot(fresh);
            if (fresh) {
                my_node = next_id + __popcll(fmask & ((1ull << lane) - 1ull));
                os.child[my_c][my_par] = my_node;  // keeps the parent's child ids canonical (copied in F2)
                const int pnode = os.node[my_par];
                backptr[my_node] = (pnode << 2) | my_c;
                ((int*)&childtab[pnode])[my_c] = my_node;
                childtab[my_node] = make_int4(0, 0, 0, 0);
            }
            next_id += __popcll(fmask);
            int4 ch = make_int4(0, 0, 0, 0);
            if (__any(reload)) {
                // a labeling that left the beam earlier and re-enters: fetch its child ids from the HBM trie
                // (written by this same wave; drain our stores, read past the L1)
                __builtin_amdgcn_s_waitcnt(0);
                if (reload) {
                    const int* cp = (const int*)&childtab[my_node];
                    ch.x = __hip_atomic_load(cp + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ch.y = __hip_atomic_load(cp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ch.z = __hip_atomic_load(cp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ch.w = __hip_atomic_load(cp + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            wave_sync();
            if (lane < nb_new) {
                const int j = d_copy[lane];
                if (j >= 0) {
                    ns.node[lane] = os.node[j];
                    ns.len[lane] = os.len[j];
                    ns.last[lane] = os.last[j];
                    ns.hist[lane] = os.hist[j];
#pragma unroll
                    for (int c = 0; c < 4; c++) ns.child[c][lane] = os.child[c][j];
                } else {
                    ns.node[lane] = my_node;
                    ns.len[lane] = os.len[my_par] + 1;
                    ns.last[lane] = my_c;
                    ns.hist[lane] = (os.hist[my_par] << 2) | (unsigned)my_c;
                    ns.child[0][lane] = ch.x;
                    ns.child[1][lane] = ch.y;
                    ns.child[2][lane] = ch.z;
                    ns.child[3][lane] = ch.w;
                }
            }
            nb = nb_new;
            cur ^= 1;
            wave_sync();
        }
    }

    // ---------------- traceback of the best labeling (slot 0 = rank 0; decode.py:207-210) --------------------
    __builtin_amdgcn_s_waitcnt(0);
    if (lane == 0) {
        const BeamState<WM>& fs = st[cur];
        int n = fs.node[0];
        const int len = fs.len[0];
        uint8_t* out = a.labels + a.label_off[seq];
        for (int p = len - 1; p >= 0; p--) {
            const int bp = __hip_atomic_load(&backptr[n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            out[p] = (uint8_t)(bp & 3);
            n = bp >> 2;
        }
        a.label_len[seq] = len;
        if (a.best_score) a.best_score[seq] = fs.ptot[0];
    }
}

template <typename PT, int R>
int launch_r(hipStream_t st, const DecodeArgs& a, int n_seq, bool lm)
{
    if (lm)
        hipLaunchKernelGGL((beam_search_kernel<PT, R, true>), dim3(n_seq), dim3(64), 0, st, a);
    else
        hipLaunchKernelGGL((beam_search_kernel<PT, R, false>), dim3(n_seq), dim3(64), 0, st, a);
    RD_HIP(hipGetLastError());
    return RD_OK;
}

template <typename PT>
int launch_pt(hipStream_t st, const DecodeArgs& a, int n_seq, bool lm)
{
    if (a.W <= Cfg<1>::WM) return launch_r<PT, 1>(st, a, n_seq, lm);
    if (a.W <= Cfg<2>::WM) return launch_r<PT, 2>(st, a, n_seq, lm);
    return launch_r<PT, 4>(st, a, n_seq, lm);
}

__global__ void lm_gate_kernel(const double* __restrict__ entropy, size_t n, double r_thr, uint32_t* __restrict__ bits)
{
    // one thread per 32 contexts
    size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t base = w * 32;
    if (base >= n) return;
    uint32_t b = 0;
    for (int i = 0; i < 32 && base + i < n; i++)
        if (entropy[base + i] < r_thr) b |= (1u << i);
    bits[w] = b;
}

}  // namespace

extern "C" int rd_decode_max_width(void) { return Cfg<4>::WM; }

// Per-context gate bits: bit = (entropy(lm[ctx]) < r_threshold)   decode.py:85-93.
// The entropies were computed once at rd_load_lm (glibc log, like the reference's math.log) and live in HBM.
static int ensure_lm_gate(rd_ctx* ctx, double r_thr)
{
    LM& lm = ctx->lm;
    if (lm.gate_valid && lm.gate_r_thr == r_thr) return RD_OK;
    const size_t n = (size_t)1 << (2 * lm.k);
    const size_t words = (n + 31) / 32;
    if (lm.gate_storage.reserve(words * 4)) return RD_ERR_NOMEM;
    lm.gate_bits = (uint32_t*)lm.gate_storage.p;
    const int threads = 256;
    const int blocks = (int)((words + threads - 1) / threads);
    hipLaunchKernelGGL(lm_gate_kernel, dim3(blocks), dim3(threads), 0, ctx->stream, lm.d_entropy, n, r_thr, lm.gate_bits);
    RD_HIP(hipGetLastError());
    lm.gate_valid = true;
    lm.gate_r_thr = r_thr;
    return RD_OK;
}

int rd_decode_dev(rd_ctx* ctx, const void* d_probs, int is_f64, const int64_t* d_seq_off, const int32_t* d_seq_len,
                  const int64_t* d_node_off, const int64_t* d_label_off, int n_seq, int64_t total_nodes, int W, int use_lm,
                  double s_thr, double r_thr, uint8_t* d_labels, int32_t* d_label_len, double* d_best_score, hipStream_t stream,
                  const int64_t* d_seq_off2, const int32_t* d_seq_split)
{
    hipStream_t st = stream ? stream : ctx->stream;
    RD_REQUIRE(W >= 1 && W <= Cfg<4>::WM, "beam_width %d out of range [1,%d]", W, Cfg<4>::WM);
    if (n_seq == 0) return RD_OK;
    if (use_lm) {
        if (!ctx->lm.loaded) {
            rd_set_error("decode with use_lm=1 but no LM table loaded (rd_load_lm)");
            return RD_ERR_STATE;
        }
        int rc = ensure_lm_gate(ctx, r_thr);
        if (rc) return rc;
    }
    if (ctx->ws_nodes_child.reserve((size_t)total_nodes * sizeof(int4))) return RD_ERR_NOMEM;
    if (ctx->ws_nodes_back.reserve((size_t)total_nodes * sizeof(int))) return RD_ERR_NOMEM;
    DecodeArgs a;
    a.probs = d_probs;
    a.seq_off = d_seq_off;
    a.seq_off2 = d_seq_off2;
    a.seq_split = d_seq_split;
    a.seq_len = d_seq_len;
    a.node_off = d_node_off;
    a.label_off = d_label_off;
    a.W = W;
    a.lm_table = use_lm ? ctx->lm.table : nullptr;
    a.lm_gate = use_lm ? ctx->lm.gate_bits : nullptr;
    a.k = use_lm ? ctx->lm.k : 0;
    a.s_thr = s_thr;
    a.childtab = ctx->ws_nodes_child.as<int4>();
    a.backptr = ctx->ws_nodes_back.as<int>();
    a.labels = d_labels;
    a.label_len = d_label_len;
    a.best_score = d_best_score;
    KernelTimer& tm = ctx->timer_decode;
    if (tm.enabled && tm.used < tm.starts.size()) RD_HIP(hipEventRecord(tm.starts[tm.used], st));
    int rc = is_f64 ? launch_pt<double>(st, a, n_seq, use_lm != 0) : launch_pt<float>(st, a, n_seq, use_lm != 0);
    if (rc) return rc;
    if (tm.enabled && tm.used < tm.starts.size()) {
        RD_HIP(hipEventRecord(tm.stops[tm.used], st));
        tm.used++;
    }
    return RD_OK;
}
