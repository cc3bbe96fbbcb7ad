// decode.hip -- CTC prefix beam search with the k-mer RNA-LM gate, one workgroup of 1-10 wavefronts per sequence (beam widths up to 256).
//
// Replaces radian/decode.py:100-212 (beam_search) and :42-96 (LM gate) of the reference.
//
// Mapping to gfx950: a sequence is a T-long serial dependency chain, so each sequence gets ONE small
// workgroup (a single 64-lane wave for W <= 12; many workgroups per CU hide each other's latency).
// Within a time step the 5*W candidate entries (for each kept beam: its copy + 4 extensions,
// decode.py:150-201) are spread over threads: candidate q = 5*i + k lives in slot q / (64 NW) of thread
// q % (64 NW) (NW waves, R slots per lane; see launch_pt for the shapes).  Candidate order q is exactly
// the reference's dict insertion order, which is what Python's stable sort falls back to on ties (decode.py:38).
//
// Exact labeling identity (the dict keyed by tuples, decode.py:171-201) is kept with a per-sequence
// trie whose node ids are canonical: every kept beam carries the ids of its four children
// (0 = never created); a new id is allocated only when a never-created child enters the top W, and
// the trie is written through to HBM so that a labeling that leaves the beam and later re-enters is
// found again under the same id.  "copy of beam j" and "extension of beam i by c" collide iff
// child[c][i] == node[j]; the two probabilities are then combined with logaddexp exactly as the
// reference does.  Back-pointers (parent<<2 | label) in HBM give the final labeling by traceback.
//
// Scores are float64 log-probabilities.  Default arithmetic (GX): log / exp / log1p evaluated with glibc 2.35's operation
// sequence (glibc_math.h), so scores are the reference's bit for bit on an x86-64 FMA host.  Fast arithmetic
// (rd_set_decode_math 0): log is ROCm's device function and logaddexp runs on the range-specific exp / log1p routines below
// (all <= 1 ulp from glibc's): scores agree to a few ulp and the labeling is identical unless two beams tie within that distance.
#include "common.h"
#include "glibc_math.h"
#include "glibc_tables.h"

#include <math.h>
#include <stdlib.h>

#include "decode_common.h"

namespace {

// One kept labeling, 64 B in LDS: a candidate lane fetches its parent with three 16-B reads.  (The 64-B stride makes the
// 4-B / 8-B reads of one field of beams i and i + 2 / i + 4 share banks -- 27 % of the kernel's LDS cycles are conflict
// cycles -- but padding the record to 80 B bought nothing at 512 sequences and cost residency at W = 25: the kernel is
// bound by instruction issue and dependent-operation latency, not by the LDS.)
struct __attribute__((aligned(16))) Beam {
    double ptot, pb;        //  0: pr_total, pr_blank          (log)
    double pnb;             // 16: pr_non_blank
    int last, len;          // 24: last label (-1: empty labeling), labeling length
    int node;               // 32: canonical trie id
    unsigned hist;          // 36: last 16 labels, 2 bits each (LM context) -- or the context hash (long contexts)
    unsigned hprev;         // 40: long contexts: hash of the context that excludes the last label
    int pad1;
    int child[4];           // 48: trie ids of the four children (0 = never created)
};
static_assert(sizeof(Beam) == 64, "Beam is one 64-B LDS record");

template <int R, int NW>
struct Cfg {
    static constexpr int NC = 64 * R * NW;  // candidate capacity
    static constexpr int WM = NC / 5;       // max beam width
};
// four waves, two candidates per lane: 512 candidates would carry 102 beams, but a beam set is built by the lanes of ONE wave (the trie
// phase), so this shape ends at 64 beams -- widths 52 ... 64 (round 5: 64 is the width people type after 32)
template <>
struct Cfg<2, 4> {
    static constexpr int WM = 64;
};
// five waves, two candidates per lane: 640 candidates = 128 beams (round 6: widths 65 ... 128, until then decode_wide.hip's at 7x the step
// time).  The per-beam work that the shapes above do with "lane = beam" runs here over TWO halves of the beam set (beam = lane + 64 h):
// the table self-check and the claim check on every wave, the trie phase on wave 0 one half after the other.
// Ten waves carry 256 beams the same way in four parts (Cfg<2, 10>: 1280 candidates; beam indices in the scatter word take 9 bits there).


// Hand-off between the waves of a sequence's workgroup.  One wave: see wave_sync.  Several waves: the LDS operations of
// this wave have completed (lgkmcnt) and every wave has arrived -- again without vmcnt(0), so the trie stores stay in flight.
// Every branch that contains a hand-off is taken by all waves or by none (its condition is computed from LDS data that
// all waves read after the same hand-off).
template <int NW>
__device__ __forceinline__ void seq_sync()
{
    if constexpr (NW == 1) {
        wave_sync();
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// node id -> beam slot: a table indexed by the low bits of the id; an entry is the rest of the id << 8 | the slot of the kept
// beam that carries it.  Every kept beam enters itself every step (its slot changes), a labeling that is not kept
// removes its entry, so a matching id is a kept labeling and one LDS round trip answers "is this extension already kept?".
// Two kept beams whose ids collide in the table are detected (each beam checks that it finds itself) and that step falls
// back to comparing against every beam.
//
// Candidate q = 5*i + k of a step lives in slot s of thread tid with q = s * (64 NW) + tid: NW waves of one workgroup
// share a sequence (W <= 12: one wave; W <= 25: two; W <= 51: four), each with R slots per lane (R = 1 in the product;
// R > 1 is the single-wave form of round 2's first half, kept for A/B builds).
template <typename PT, int R, int NW, bool LM, bool HC, bool GX>
__device__ __forceinline__ void beam_search_body(const DecodeArgs& a, const int seq)
{
    static_assert(LM || !HC, "hashed contexts only exist with an LM");
    constexpr int WM = Cfg<R, NW>::WM;
    constexpr int TPB = 64 * NW;
    constexpr int NS = R * NW;          // key segments: one per (slot, wave), in insertion order
    // keys per pass of the ranking loop (all their LDS reads are in flight together): 16 shortens a lone wave's step (one
    // round trip for <= 16 survivors); with R slots per lane the 32 key registers would cost residency, which is what the
    // R > 1 form is for (many sequences)
    constexpr int KG = R == 1 ? 16 : 4;
    constexpr int SEG = 64 + KG;        // doubles per segment: 64 keys + the padding of the last group
    constexpr int HB = (WM + 63) / 64;  // 64-beam parts of the beam set: per-beam lanes handle beam lane + 64 h
    constexpr int LOG_TN = WM > 128 ? 12 : WM > 64 ? 11 : R * NW <= 2 ? 9 : 10;
    constexpr int TN = 1 << LOG_TN;     // (LDS per sequence bounds the resident waves: 2 KiB / 4 KiB; 8 / 16 KiB for 128 / 256 beams)
    constexpr int SB = WM > 128 ? 9 : 8;            // bits of a beam index inside d_sel (all ones = none)
    constexpr int SMASK = (1 << SB) - 1;
    static_assert(WM <= 64 * HB && HB <= 4 && WM <= 256, "the kept beams fit the lanes of one wave, up to four times; a table entry holds a slot in 8 bits");
    static_assert(HB == 1 || !HC, "hashed contexts exist for up to 64 beams");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = NW == 1 ? 0 : __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = a.seq_len[seq];
    const PT* __restrict__ probs = (const PT*)a.probs;
    const int64_t row_a = a.seq_off[seq];
    const int64_t row_b = a.seq_off2 ? a.seq_off2[seq] : row_a;
    const int split = a.seq_off2 ? a.seq_split[seq] : 0;
    int4* __restrict__ childtab = a.childtab + a.node_off[seq];
    int* __restrict__ backptr = a.backptr + a.node_off[seq];
    const int W = a.W;
    const unsigned ctx_mask = LM ? ((a.k >= 16) ? 0xffffffffu : ((1u << (2 * a.k)) - 1u)) : 0u;

    __shared__ Beam st[2][WM];
    // per-beam scratch of the merge phases (5 x WM doubles) and the ranking's key segments.  One wave: the keys overlay the
    // scratch (they are written after its last read in program order, and LDS operations of a wave execute in order); the
    // footprint decides how many sequences stay resident per CU.
    constexpr int SCR = 5 * WM, KEYS = NS * SEG;
    __shared__ __attribute__((aligned(16))) double dbuf[NW == 1 ? (SCR > KEYS ? SCR : KEYS) : SCR + KEYS + (SCR & 1)];
    double* const cpy_pnb = dbuf;
    double* const cpy_pb = dbuf + WM;
    double* const mb_v = dbuf + 2 * WM;
    double* const mP = dbuf + 3 * WM;
    double* const mQ = dbuf + 4 * WM;
    double* const keyC = NW == 1 ? dbuf : dbuf + SCR + (SCR & 1);   // 16-B aligned: keys of the ranking's survivors, compacted per segment
    __shared__ int mb_q[WM];
    __shared__ int d_sel[WM], newslot[WM];   // d_sel: who fills new slot r = (copied beam or all ones) | parent beam << SB | (1 + label) << 2 SB  (SB = 8; 9 for 256 beams)
    __shared__ unsigned claims[WM];
    __shared__ double lp[64][5];
    __shared__ double praw[LM ? 64 : 1][5];
    __shared__ double sent[LM ? 64 : 1];
    __shared__ int seg_s[NS], seg_v[NS], mflag[NW];
    __shared__ unsigned tab[TN];
    // glibc arithmetic: exp's table.  One candidate per lane (the forms whose point is a short time step): a copy in LDS, 2 KiB.
    // Two candidates per lane (the form for thousands of sequences, where LDS per sequence decides how many stay resident --
    // 16 per CU for a 4096-sequence launch -- and other waves hide a global load's latency): read in place, through the L1.
    constexpr bool GXL = GX && R == 1;
    __shared__ uint64_t gx_lds[GXL ? 256 : 1];
    if constexpr (GXL)
        for (int i = tid; i < 256; i += TPB) gx_lds[i] = g_gm_exp_tab[i];
    const uint64_t* const gx_exp = GXL ? gx_lds : g_gm_exp_tab;
    __shared__ __attribute__((aligned(16))) unsigned ring[2][HC ? WM : 1][16];   // long contexts: the last 256 labels of each beam, 2 bits each

    for (int i = tid; i < TN; i += TPB) tab[i] = i == 0 ? 0u : 0xffffffffu;    // (the empty labeling: id 0 in slot 0)
    // decode.py:128-132: the empty labeling with pr_blank = pr_total = log(1)
    if (tid == 0) {
        Beam& b = st[0][0];
        b.ptot = 0.0;
        b.pb = 0.0;
        b.pnb = -INFINITY;
        b.last = -1;
        b.len = 0;
        b.node = 0;
        b.hist = 0u;
        b.hprev = 0u;
        b.pad1 = 0;
        for (int c = 0; c < 4; c++) b.child[c] = 0;
        childtab[0] = make_int4(0, 0, 0, 0);
        backptr[0] = 0;
    }
    int nb = 1;              // beams currently kept (workgroup-uniform)
    int next_id = 1;         // next free trie node id (wave 0)
    int cur = 0;
    bool missed = false;     // sparse LM: a labeling this lane built has a context the model does not hold
    seq_sync<NW>();

    for (int t0 = 0; t0 < T; t0 += 64) {
        // ---- per-tile prepass: one lane per time step computes the 5 log-probabilities (decode.py:165,168,193,195
        //      take math.log of mat[t][c]) and, with an LM, the entropy of the renormalised base distribution
        //      (decode.py:135-138).
        if (NW == 1 || wv == 0) {
            const int t = t0 + lane;
            if (t < T) {
                // (loops of one class each, NOT unrolled: one inlined copy of log per loop instead of nine -- the copies are
                // what the compiler spills scalars around, and the prepass is 1/64 of the work)
                const PT* __restrict__ prow = probs + ((t < split ? row_a : row_b) + t) * 5;
                double s4 = 0.0;      // ((p0 + p1) + p2) + p3, the order normalise() adds in
#pragma unroll 1
                for (int c = 0; c < 5; c++) {
                    const double pc = (double)prow[c];
                    lp[lane][c] = safe_log<GX>(pc);
                    if constexpr (LM) {
                        praw[lane][c] = pc;
                        if (c < 4) s4 = c == 0 ? pc : s4 + pc;
                    }
                }
                if constexpr (LM) {
                    // normalise(): sum(dist) in float64 (numpy-1.19 semantics); float32 rows divide in float32
                    const double s = s4;
                    double ent = 0.0;
                    bool any = false;
#pragma unroll 1
                    for (int c = 0; c < 4; c++) {
                        const double pc = praw[lane][c];
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
                    sent[lane] = any ? -ent : 0.0;
                }
            }
        }
        seq_sync<NW>();

        const int tend = (T - t0) < 64 ? (T - t0) : 64;
        for (int tt = 0; tt < tend; tt++) {
            const Beam* os = st[cur];                 // (no __restrict__: the scores-only step below writes these records in place)
            Beam* __restrict__ ns = st[cur ^ 1];
            const int ncand = 5 * nb;

            // ---------------- Phase A: candidate scores; which extension equals which kept labeling? -------------
            // A time step is a chain of LDS round trips; the reads are written unconditionally (clamped indices, results
            // selected afterwards, no && / || over loads) so that each phase issues ONE batch of independent reads and
            // waits once -- as conditional reads hipcc emits a wait and a branch per read.
            bool valid[R], is_copy[R];
            int bi[R], kk[R], pj[R], dcopy[R], xch[R];
            double c_ptot[R], c_pnb[R], c_pb[R];
            // batch 1: the parent records, log p(blank), the weakest kept beam (ranking threshold), this lane's kept beam
            const double lp_blank = lp[tt][4];
            // the signal side of the LM gate (decode.py:91) is a property of the time step: closed, no candidate looks at the LM
            bool s_open = false;
            if constexpr (LM) s_open = __builtin_amdgcn_readfirstlane((int)(sent[tt] > a.s_thr)) != 0;
            const double ptot_last = os[nb - 1].ptot;
            int myn[HB];
            if constexpr (HB == 1) {
                myn[0] = os[lane < nb ? lane : 0].node;
            } else {
#pragma unroll
                for (int h = 0; h < HB; h++) myn[h] = os[lane + 64 * h < nb ? lane + 64 * h : 0].node;
            }
            double2 pp[R];
            double pnb_i[R];
            int2 ll[R];
            int chx[R];
#pragma unroll
            for (int s = 0; s < R; s++) {
                const int q = s * TPB + tid;
                valid[s] = q < ncand;
                const int i = valid[s] ? q / 5 : 0;
                const int k = q - 5 * (q / 5);
                bi[s] = i;
                kk[s] = k;
                is_copy[s] = k == 0;
                pp[s] = *(const double2*)&os[i].ptot;               // pr_total, pr_blank
                pnb_i[s] = os[i].pnb;
                ll[s] = *(const int2*)&os[i].last;                   // last, len
                // (all four child ids from the record's base address and a select: as a 4-B read at a computed address hipcc
                // issues it after the wait for the reads above -- one more round trip)
                int4 c4 = *(const int4*)&os[i].child[0];
                asm("" : "+v"(c4.x), "+v"(c4.y), "+v"(c4.z), "+v"(c4.w));   // (opaque: or the select below becomes four conditional reads)
                const int ci = (k - 1) & 3;
                const int c_lo = (ci & 1) ? c4.y : c4.x, c_hi = (ci & 1) ? c4.w : c4.z;
                chx[s] = (ci & 2) ? c_hi : c_lo;
            }
            if (tid < W) claims[tid] = 0u;
            // batch 2: log p(label), the id-table probe of the extension's child id, and every kept beam looks itself up: the
            // table resolves all kept beams iff each finds itself (every wave checks all beams)
            unsigned my_e[HB];
            if constexpr (HB == 1) {
                my_e[0] = tab[myn[0] & (TN - 1)];
            } else {
#pragma unroll
                for (int h = 0; h < HB; h++) my_e[h] = tab[myn[h] & (TN - 1)];
            }
            double lpc[R];
            unsigned p_e[R];
#pragma unroll
            for (int s = 0; s < R; s++) {
                const int c = is_copy[s] ? ll[s].x : kk[s] - 1;     // label whose probability this candidate consumes
                lpc[s] = lp[tt][c < 0 ? 0 : c];
                lpc[s] = c < 0 ? -INFINITY : lpc[s];
                const int x = (valid[s] & !is_copy[s]) ? chx[s] : 0;
                xch[s] = x;
                p_e[s] = tab[x & (TN - 1)];
            }
            bool tab_ok;
            if constexpr (HB == 1) {
                tab_ok = !__any((lane < nb) & (my_e[0] != ((((unsigned)myn[0] >> LOG_TN) << 8) | (unsigned)lane)));
            } else {
                bool tab_bad = false;
#pragma unroll
                for (int h = 0; h < HB; h++) tab_bad |= (lane + 64 * h < nb) & (my_e[h] != ((((unsigned)myn[h] >> LOG_TN) << 8) | (unsigned)(lane + 64 * h)));
                tab_ok = !__any(tab_bad);
            }
#pragma unroll
            for (int s = 0; s < R; s++) {
                const int i = bi[s], k = kk[s];
                const int last_i = ll[s].x;
                if constexpr (LM) {
                    // decode.py:157-163 (copy: context excludes the last label) and :180-184 (extend)
                    const int len_i = ll[s].y;
                    const int c = is_copy[s] ? last_i : k - 1;
                    const int need = is_copy[s] ? a.k + 1 : a.k;
                    if (s_open && valid[s] && c >= 0 && len_i >= need) {
                        unsigned ctx;
                        if constexpr (HC) {
                            ctx = (is_copy[s] ? os[i].hprev : os[i].hist) & a.tmask;
                        } else {
                            const unsigned h = os[i].hist;
                            ctx = (is_copy[s] ? (h >> 2) : h) & ctx_mask;
                        }
                        const bool gate = (a.lm_gate[ctx >> 5] >> (ctx & 31)) & 1u;
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
                            lpc[s] = safe_log<GX>(val);
                        }
                    }
                }
                // COPY decode.py:150-175 / EXTEND decode.py:186-201
                const double pnb_c = (last_i >= 0) ? pnb_i[s] + lpc[s] : -INFINITY;
                const double pb_c = pp[s].x + lp_blank;
                const double v = ((last_i == k - 1) ? pp[s].y : pp[s].x) + lpc[s];
                c_pnb[s] = is_copy[s] ? pnb_c : v;
                c_pb[s] = is_copy[s] ? pb_c : -INFINITY;
                c_ptot[s] = is_copy[s] ? 0.0 : v;            // copies: lae(pb_c, pnb_c) below
                dcopy[s] = is_copy[s] ? i : -1;
                if (valid[s] & is_copy[s]) {
                    cpy_pnb[i] = pnb_c;
                    cpy_pb[i] = pb_c;
                    mb_q[i] = -1;
                    newslot[i] = -1;
                }
                // the labeling "beam i + label" is already kept iff its trie id is in the id table
                pj[s] = (tab_ok & (xch[s] != 0) & ((p_e[s] >> 8) == ((unsigned)xch[s] >> LOG_TN))) ? (int)(p_e[s] & 0xffu) : -1;
            }
            if (!tab_ok) {   // two kept beams share a table entry (rare): compare against every beam
                for (int j = 0; j < nb; j++) {
                    const int nj = os[j].node;
#pragma unroll
                    for (int s = 0; s < R; s++)
                        if (xch[s] != 0 && nj == xch[s]) pj[s] = j;
                }
            }
            // does any extension of the step merge into a kept labeling?  (workgroup-uniform: the branch below holds hand-offs)
            bool any_merge = false;
#pragma unroll
            for (int s = 0; s < R; s++) any_merge |= __any(pj[s] >= 0);
            if constexpr (NW > 1) {
                if (lane == 0) mflag[wv] = any_merge ? 1 : 0;
            }
            seq_sync<NW>();
            if constexpr (NW > 1) {
                int m = 0;
#pragma unroll
                for (int w = 0; w < NW; w++) m |= mflag[w];
                any_merge = __builtin_amdgcn_readfirstlane(m) != 0;
            }

            // ---------------- lae pass 1: copies: pr_total = logaddexp(pr_blank, pr_non_blank) (decode.py:174-175);
            //                  merging extensions: pr_non_blank of the merged entry = logaddexp(copy.pnb, v) (decode.py:199)
#pragma unroll
            for (int s = 0; s < R; s++) {
                const bool mext = pj[s] >= 0;
                const double cp = cpy_pnb[mext ? pj[s] : 0];
                const double x = mext ? cp : c_pb[s];
                const double y = mext ? c_ptot[s] : c_pnb[s];
                const double r = GX ? lae_gx(x, y, gx_exp) : lae(x, y);
                c_ptot[s] = is_copy[s] ? r : c_ptot[s];
                if (mext) {
                    mb_q[pj[s]] = s * TPB + tid;
                    mb_v[pj[s]] = c_ptot[s];
                    mQ[pj[s]] = r;
                }
            }
            if (any_merge) {
                seq_sync<NW>();
                // ---------- lae pass 2: pr_total of the merged entry = logaddexp(copy.total, v) (decode.py:200-201), by the copy's lane
#pragma unroll
                for (int s = 0; s < R; s++) {
                    const int qe = mb_q[bi[s]];
                    const double mv = mb_v[bi[s]];
                    const bool m = valid[s] & is_copy[s] & (qe >= 0);
                    const double r = GX ? lae_gx(c_ptot[s], m ? mv : -INFINITY, gx_exp) : lae(c_ptot[s], m ? mv : -INFINITY);
                    if (m) mP[bi[s]] = r;
                }
                seq_sync<NW>();
                // the merged entry lives in whichever of the two candidates was inserted first (dict order); the other is gone
#pragma unroll
                for (int s = 0; s < R; s++) {
                    const int q = s * TPB + tid;
                    const int j = is_copy[s] ? bi[s] : (pj[s] >= 0 ? pj[s] : 0);      // the copy's beam
                    const int qe = mb_q[j];
                    const double P = mP[j], Q = mQ[j], cb = cpy_pb[j];
                    const bool merged = valid[s] & (is_copy[s] ? qe >= 0 : pj[s] >= 0);
                    const int qother = is_copy[s] ? qe : 5 * j;
                    const bool mk = merged & (q < qother);                           // merged and inserted first: holds the entry
                    const bool mke = mk & !is_copy[s];
                    c_ptot[s] = mk ? P : c_ptot[s];
                    c_pnb[s] = mk ? Q : c_pnb[s];
                    c_pb[s] = mke ? cb : c_pb[s];
                    dcopy[s] = mke ? j : dcopy[s];
                    valid[s] = valid[s] & (!merged | mk);
                }
            }

            // ---------------- Phase D: rank by (pr_total desc, insertion order asc)  decode.py:35-39,145 ------
            // Only candidates that can reach the top W are ranked.  Every kept labeling survives this step as an entry
            // (its copy, merged or not) whose pr_total >= pr_total_old + log p(blank) >= tau := the weakest kept beam's
            // pr_total + log p(blank): with W beams kept there are W entries >= tau, so an entry below tau is not among
            // the best W.  The survivors' keys are compacted into LDS in insertion order -- one segment per (slot, wave),
            // segments in order -- and each survivor counts the keys ahead of it (16-B LDS broadcasts, two keys per read).
            double key[R];
            bool surv[R];
            int lidx[R], scnt[R], vcnt[R];
            const double tau = (nb == W) ? ptot_last + lp_blank : -INFINITY;
#pragma unroll
            for (int s = 0; s < R; s++) {
                key[s] = valid[s] ? c_ptot[s] : __builtin_nan("");
                surv[s] = valid[s] && key[s] >= tau;
                const unsigned long long mv = __ballot(valid[s]), ms = __ballot(surv[s]);
                vcnt[s] = __popcll(mv);
                scnt[s] = __popcll(ms);
                lidx[s] = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(ms >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ms, 0u));
                double* kseg = keyC + (s * NW + wv) * SEG;
                if (lane < KG) kseg[scnt[s] + lane] = -INFINITY;  // padding of the last group
                if (surv[s]) kseg[lidx[s]] = key[s];
                if constexpr (NW > 1) {
                    if (lane == 0) {
                        seg_s[s * NW + wv] = scnt[s];
                        seg_v[s * NW + wv] = vcnt[s];
                    }
                }
            }
            seq_sync<NW>();
            int segn[NS];     // survivors per segment (scalars)
            int nvalid = 0;
#pragma unroll
            for (int g = 0; g < NS; g++) {
                if constexpr (NW > 1) {
                    segn[g] = __builtin_amdgcn_readfirstlane(seg_s[g]);
                    nvalid += __builtin_amdgcn_readfirstlane(seg_v[g]);
                } else {
                    segn[g] = scnt[g];
                    nvalid += vcnt[g];
                }
            }
            int rank[R];
#pragma unroll
            for (int s = 0; s < R; s++) rank[s] = 0;
#pragma unroll
            for (int g = 0; g < NS; g++) {
                const double* kseg = keyC + g * SEG;
                for (int j = 0; j < segn[g]; j += KG) {
                    double2 kq[KG / 2];
#pragma unroll
                    for (int u = 0; u < KG / 2; u++) kq[u] = *(const double2*)&kseg[j + 2 * u];
#pragma unroll
                    for (int u = 0; u < KG / 4; u++)
#pragma unroll
                        for (int s = 0; s < R; s++)
                            rank[s] = count4_gt(rank[s], kq[2 * u].x, kq[2 * u].y, kq[2 * u + 1].x, kq[2 * u + 1].y, key[s]);
                }
            }
            const int nb_new = nvalid < W ? nvalid : W;

            // ---------------- the step that changes nothing but the scores (one wave, one candidate per lane) ----------------
            // When every entry of the new top W is a kept labeling -- its copy, or its copy merged with an extension -- at the rank of
            // its old slot, the beam set, its order, every trie id and every id-table entry stay as they are: the three scores go into
            // the records where they lie, and the scatter, the claim check and the trie phase (a quarter of the step) are skipped.
            // On peaked rows that is almost every step (counted on the CPU over the bench's windows: 97 % of the steps at W = 10; on soft rows 1 %).
            // Conservative on ties: the count is of strictly greater keys, so two equal keys share a rank, one of them misses its
            // slot's number and the step takes the general path, which orders ties as the reference does.
            if constexpr (NW == 1 && R == 1) {
                const bool top = surv[0] && rank[0] < W;
                const unsigned long long m_top = __ballot(top), m_same = __ballot(top && dcopy[0] >= 0 && rank[0] == dcopy[0]);
                if (m_top == m_same && __popcll(m_top) == nb && nb_new == nb) {
                    if (top) {
                        Beam* const here = st[cur] + rank[0];      // (every read of the old records lies before the hand-off after the keys)
                        *(double2*)&here->ptot = make_double2(c_ptot[0], c_pb[0]);
                        here->pnb = c_pnb[0];
                    }
                    wave_sync();
                    continue;
                }
            }

            // ---------------- Phase E: the kept candidates move to their new beam slot ------------------------
            // The count above is of strictly greater keys: equal keys get the same count, so a slot below nb_new that is
            // not claimed exactly once means a tie that matters; only then is the count redone with the insertion-order
            // rule (exact 0 probabilities make such ties; softmax rows do not) and the slots are written again.
            auto scatter = [&]() {
#pragma unroll
                for (int s = 0; s < R; s++) {
                    if (surv[s] && rank[s] < W) {
                        const int r = rank[s];
                        *(double2*)&ns[r].ptot = make_double2(c_ptot[s], c_pb[s]);
                        ns[r].pnb = c_pnb[s];
                        d_sel[r] = (dcopy[s] & SMASK) | (bi[s] << SB) | (kk[s] << (2 * SB));
                        atomicAdd(&claims[r], 1u);
                        if (dcopy[s] >= 0) newslot[dcopy[s]] = r;
                    }
                }
            };
            scatter();
            seq_sync<NW>();
            // one batch: the claim counts, Phase F's input, and what an old beam needs to know to leave the id table
            unsigned n_claims[HB], my_tn2[HB];
            int sel[HB], my_newslot[HB];
            bool claim_bad;
            if constexpr (HB == 1) {
                n_claims[0] = claims[lane < nb_new ? lane : 0];
                sel[0] = d_sel[lane < nb_new ? lane : 0];
                my_newslot[0] = newslot[lane < nb ? lane : 0];
                my_tn2[0] = tab[myn[0] & (TN - 1)];
                asm volatile("" : "+v"(sel[0]), "+v"(my_newslot[0]), "+v"(my_tn2[0]));   // (all four requested before the tie branch)
                claim_bad = (lane < nb_new) & (n_claims[0] != 1u);
            } else {
                claim_bad = false;
#pragma unroll
                for (int h = 0; h < HB; h++) {
                    const int bl = lane + 64 * h;
                    n_claims[h] = claims[bl < nb_new ? bl : 0];
                    sel[h] = d_sel[bl < nb_new ? bl : 0];
                    my_newslot[h] = newslot[bl < nb ? bl : 0];
                    my_tn2[h] = tab[myn[h] & (TN - 1)];
                    asm volatile("" : "+v"(sel[h]), "+v"(my_newslot[h]), "+v"(my_tn2[h]));
                    claim_bad |= (bl < nb_new) & (n_claims[h] != 1u);
                }
            }
            if (__any(claim_bad)) {      // (every wave looks at all slots: workgroup-uniform)
#pragma unroll
                for (int s = 0; s < R; s++) rank[s] = 0;
#pragma unroll
                for (int g = 0; g < NS; g++) {
                    const double* kseg = keyC + g * SEG;
                    for (int j = 0; j < segn[g]; j++) {
                        const double kv = kseg[j];
#pragma unroll
                        for (int s = 0; s < R; s++) {
                            // insertion order = (segment, index in segment); this candidate's segment is s * NW + wv
                            const bool before = g < s * NW + wv || (g == s * NW + wv && j < lidx[s]);
                            rank[s] += ((kv > key[s]) || (kv == key[s] && before)) ? 1 : 0;
                        }
                    }
                }
                if (tid < nb) newslot[tid] = -1;      // the first attempt may have placed a labeling that is not kept after all
                if constexpr (NW > 1) seq_sync<NW>();
                scatter();
                seq_sync<NW>();
#pragma unroll
                for (int h = 0; h < HB; h++) {
                    sel[h] = d_sel[lane + 64 * h < nb_new ? lane + 64 * h : 0];
                    my_newslot[h] = newslot[lane + 64 * h < nb ? lane + 64 * h : 0];
                }
            }

            // ---------------- Phase F: the new beam set: trie ids, labeling state (wave 0) ----------------------
            // Lane r < nb_new builds beam r.  A copy takes its record from the old beam; an extension takes its parent's,
            // appends its label and gets its canonical id: the parent's child id if that child was ever created (its own
            // child ids then come back from the HBM trie), else a fresh id.  A fresh id is also patched into the parent's
            // NEW record when the parent is kept (newslot), so that child ids stay canonical.
            if constexpr (HB == 1) {
            if (NW == 1 || wv == 0) {
                const bool act = lane < nb_new;
                const int j = act ? (sel[0] & 0xff) : 0;                // copied beam (0xff: none)
                const int par = act ? (sel[0] >> 8) & 0xff : 0;
                const int cl = act ? (sel[0] >> 16) - 1 : 0;
                const bool is_ext = act && j == 0xff;
                const int src = is_ext ? par : j;
                // one batch: the source record and the parent's new slot (for the patch at the end)
                const int4 meta = *(const int4*)&os[src].node;       // node, hist, pad, pad
                const int2 sl2 = *(const int2*)&os[src].last;        // last, len
                const int4 chs = *(const int4*)&os[src].child[0];
                const int nid_old = os[src].child[cl & 3];           // (extensions: cl = the appended label)
                const int ps = newslot[par];
                // a labeling that is not kept leaves the id table (before the new beams enter theirs: in-order LDS)
                if ((lane < nb) & (my_newslot[0] < 0) & ((my_tn2[0] >> 8) == ((unsigned)myn[0] >> LOG_TN))) tab[myn[0] & (TN - 1)] = 0xffffffffu;
                const bool fresh = is_ext && nid_old == 0;
                const bool reload = is_ext && nid_old != 0;
                const unsigned long long fmask = __ballot(fresh);
                const int my_node = fresh ? next_id + __popcll(fmask & ((1ull << lane) - 1ull)) : nid_old;
                if (fresh) {
                    backptr[my_node] = (meta.x << 2) | cl;
                    ((int*)&childtab[meta.x])[cl] = my_node;
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
                    // the loads have landed before the block ends: no load is pending at the loop's back edge, so hipcc puts
                    // no vmcnt wait (which would also wait for the trie STORES of the step) into the next step
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(ch.x), "+v"(ch.y), "+v"(ch.z), "+v"(ch.w));
                }
                if (act) {
                    const int new_node = is_ext ? my_node : meta.x;
                    *(int2*)&ns[lane].last = make_int2(is_ext ? cl : sl2.x, is_ext ? sl2.y + 1 : sl2.y);
                    unsigned h_new = ((unsigned)meta.y << 2) | (unsigned)cl, hp_new = (unsigned)meta.z;
                    if constexpr (HC) {
                        // the beam's label ring moves with it; an extension appends its label at position len (mod 256) and the
                        // label k positions back leaves the hash window
                        const uint4* rs = (const uint4*)ring[cur][src];
                        uint4* rd = (uint4*)ring[cur ^ 1][lane];
                        const uint4 r0 = rs[0], r1 = rs[1], r2 = rs[2], r3 = rs[3];
                        rd[0] = r0;
                        rd[1] = r1;
                        rd[2] = r2;
                        rd[3] = r3;
                        const int len_p = sl2.y;                                   // the parent's length (extensions)
                        const int pout = (len_p - a.k) & 255, pin = len_p & 255;
                        const unsigned wout = ring[cur][src][pout >> 4], win = ring[cur][src][pin >> 4];
                        const unsigned lout = len_p >= a.k ? (wout >> ((pout & 15) * 2)) & 3u : 0u;
                        if (is_ext) {
                            const int sh = (pin & 15) * 2;
                            ring[cur ^ 1][lane][pin >> 4] = (win & ~(3u << sh)) | ((unsigned)cl << sh);
                        }
                        h_new = (unsigned)meta.y * kHashB + (unsigned)cl - lout * a.bk;
                        hp_new = is_ext ? (unsigned)meta.y : (unsigned)meta.z;
                    }
                    if constexpr (LM && !HC) {
                        if (a.lm_missing && is_ext && sl2.y + 1 >= a.k && t0 + tt + 1 < T) {
                            const unsigned cx = h_new & ctx_mask;
                            missed |= ((a.lm_missing[cx >> 5] >> (cx & 31)) & 1u) != 0u;
                        }
                    }
                    *(int4*)&ns[lane].node = make_int4(new_node, is_ext ? (int)h_new : meta.y, (int)hp_new, 0);
                    *(int4*)&ns[lane].child[0] = make_int4(is_ext ? ch.x : chs.x, is_ext ? ch.y : chs.y, is_ext ? ch.z : chs.z, is_ext ? ch.w : chs.w);
                    tab[new_node & (TN - 1)] = (((unsigned)new_node >> LOG_TN) << 8) | (unsigned)lane;
                }
                // (after the record writes above in program order: LDS operations of a wave execute in order)
                if (fresh && ps >= 0) ns[ps].child[cl] = my_node;
            }
            } else {
                // 65 ... 128 beams: wave 0 builds the new beam set one half after the other (lane r of half h builds beam r + 64 h).  Same
                // order of LDS operations as above where it matters: labelings that are not kept leave the id table before ANY new beam
                // enters; a fresh child id is patched into its parent's new record after EVERY record has been written.
                if (wv == 0) {
#pragma unroll
                    for (int h = 0; h < HB; h++)
                        if ((lane + 64 * h < nb) & (my_newslot[h] < 0) & ((my_tn2[h] >> 8) == ((unsigned)myn[h] >> LOG_TN))) tab[myn[h] & (TN - 1)] = 0xffffffffu;
                    bool fresh_h[HB];
                    int ps_h[HB], cl_h[HB], node_h[HB];
#pragma unroll
                    for (int h = 0; h < HB; h++) {
                        const int bl = lane + 64 * h;
                        const bool act = bl < nb_new;
                        const int j = act ? (sel[h] & SMASK) : 0;               // copied beam (all ones: none)
                        const int par = act ? (sel[h] >> SB) & SMASK : 0;
                        const int cl = act ? (sel[h] >> (2 * SB)) - 1 : 0;
                        const bool is_ext = act && j == SMASK;
                        const int src = is_ext ? par : j;
                        const int4 meta = *(const int4*)&os[src].node;       // node, hist, pad, pad
                        const int2 sl2 = *(const int2*)&os[src].last;        // last, len
                        const int4 chs = *(const int4*)&os[src].child[0];
                        const int nid_old = os[src].child[cl & 3];           // (extensions: cl = the appended label)
                        const int ps = newslot[par];
                        const bool fresh = is_ext && nid_old == 0;
                        const bool reload = is_ext && nid_old != 0;
                        const unsigned long long fmask = __ballot(fresh);
                        const int my_node = fresh ? next_id + __popcll(fmask & ((1ull << lane) - 1ull)) : nid_old;
                        if (fresh) {
                            backptr[my_node] = (meta.x << 2) | cl;
                            ((int*)&childtab[meta.x])[cl] = my_node;
                            childtab[my_node] = make_int4(0, 0, 0, 0);
                        }
                        next_id += __popcll(fmask);
                        int4 ch = make_int4(0, 0, 0, 0);
                        if (__any(reload)) {
                            __builtin_amdgcn_s_waitcnt(0);
                            if (reload) {
                                const int* cp = (const int*)&childtab[my_node];
                                ch.x = __hip_atomic_load(cp + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                ch.y = __hip_atomic_load(cp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                ch.z = __hip_atomic_load(cp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                ch.w = __hip_atomic_load(cp + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                            asm volatile("s_waitcnt vmcnt(0)" : "+v"(ch.x), "+v"(ch.y), "+v"(ch.z), "+v"(ch.w));
                        }
                        if (act) {
                            const int new_node = is_ext ? my_node : meta.x;
                            *(int2*)&ns[bl].last = make_int2(is_ext ? cl : sl2.x, is_ext ? sl2.y + 1 : sl2.y);
                            const unsigned h_new = ((unsigned)meta.y << 2) | (unsigned)cl;
                            if constexpr (LM) {
                                if (a.lm_missing && is_ext && sl2.y + 1 >= a.k && t0 + tt + 1 < T) {
                                    const unsigned cx = h_new & ctx_mask;
                                    missed |= ((a.lm_missing[cx >> 5] >> (cx & 31)) & 1u) != 0u;
                                }
                            }
                            *(int4*)&ns[bl].node = make_int4(new_node, is_ext ? (int)h_new : meta.y, meta.z, 0);
                            *(int4*)&ns[bl].child[0] = make_int4(is_ext ? ch.x : chs.x, is_ext ? ch.y : chs.y, is_ext ? ch.z : chs.z, is_ext ? ch.w : chs.w);
                            tab[new_node & (TN - 1)] = (((unsigned)new_node >> LOG_TN) << 8) | (unsigned)bl;
                        }
                        fresh_h[h] = fresh;
                        ps_h[h] = ps;
                        cl_h[h] = cl;
                        node_h[h] = my_node;
                    }
#pragma unroll
                    for (int h = 0; h < HB; h++)
                        if (fresh_h[h] && ps_h[h] >= 0) ns[ps_h[h]].child[cl_h[h]] = node_h[h];
                }
            }
            nb = nb_new;
            cur ^= 1;
            seq_sync<NW>();
        }
    }

    // ---------------- traceback of the best labeling (slot 0 = rank 0; decode.py:207-210) --------------------
    int tid_end = tid;
    asm volatile("" : "+v"(tid_end));   // (the lane mask of "tid == 0" is computed here, not carried in SGPRs from kernel entry)
    const bool any_missed = LM && __any(missed);     // (Phase F runs on wave 0, which also holds thread 0)
    if (tid_end == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        // The output pointers are only needed here.  Read through the kernarg segment behind an opaque copy of its address, they are
        // loaded now instead of at kernel entry -- loaded there, the compiler parks them in spilled SGPRs for the whole time loop.
        const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        const DecodeArgs* ap = (const DecodeArgs*)ka;
        const Beam& fs = st[cur][0];
        int n = fs.node;
        const int len = fs.len;
        uint8_t* out = ap->labels + ap->label_off[seq];
        for (int p = len - 1; p >= 0; p--) {
            const int bp = __hip_atomic_load(&backptr[n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            out[p] = (uint8_t)(bp & 3);
            n = bp >> 2;
        }
        ap->label_len[seq] = any_missed ? -1 : len;
        if (ap->best_score) ap->best_score[seq] = fs.ptot;
    }
}


template <typename PT, int R, int NW, bool LM, bool HC, bool GX>
__global__ __launch_bounds__(64 * NW) void beam_search_kernel(DecodeArgs a)
{
    beam_search_body<PT, R, NW, LM, HC, GX>(a, (int)blockIdx.x);
}

// The same search behind a work queue: `gridDim.x` workgroups -- as many as the CUs they may run on keep RESIDENT -- take the sequences in
// index order (the caller sorts longest first) from a counter in HBM until it runs past n_seq, which every workgroup reaches.  For a group
// of the reads pipeline with more sequences than its decode partition holds (pipe_reads.hip): launched as one workgroup per sequence, the
// surplus workgroups wait in the dispatcher, and a dispatch that cannot place its workgroups blocks the other queues of its pipe -- the
// forward lanes stood still for the length of the longest chain (200-ms stalls in tools/policy_probe.py's trace; DESIGN_LOG.md round 5).
template <typename PT, int R, int NW, bool LM, bool HC, bool GX>
__global__ __launch_bounds__(64 * NW) void beam_search_queue_kernel(DecodeArgs a, int n_seq, int* counter)
{
    __shared__ int s_next;
    for (;;) {
        if (threadIdx.x == 0) s_next = atomicAdd(counter, 1);
        __syncthreads();
        const int seq = __builtin_amdgcn_readfirstlane(s_next);
        __syncthreads();                       // (everyone has read it before thread 0 draws again)
        if (seq >= n_seq) return;
        beam_search_body<PT, R, NW, LM, HC, GX>(a, seq);
        __syncthreads();                       // (the sequence's LDS state is dead before the next one initialises it)
    }
}


// ---- two sequences per wave (W <= 12) -----------------------------------------------------------------------------------
// At the reference's default width (basecall.py:32: beam 6) a step has at most 30 candidates: half of a wave.  Here a wave
// carries TWO sequences, lanes 0-31 and lanes 32-63 ("halves"), through the same step as beam_search_kernel<PT, R, 1, ...>:
// every phase is the one-wave kernel's with "wave-uniform" replaced by "uniform within the half" -- nb, next_id, the trie
// pointers, the gate's signal side live in vector registers; ballots are taken per half; a branch is taken when either half
// needs it and is a no-op for the lanes of the other; a half whose sequence has ended idles (valid = false everywhere) while
// the other finishes.  Per issued instruction twice the sequences advance: the form for launches that are bound by
// instruction issue (thousands of windows; waves beside a forward), not for a few long chains.  No hashed contexts (the
// one-wave kernel takes those).
// R = 1: W <= 6, one candidate per lane of the half.  R = 2 (round 4): 7 <= W <= 12 -- the metric's width, beam 10 -- with two
// candidates per lane (candidate q = s * 32 + lane of the half, the reference's insertion order again): the instruction stream
// of the one-wave kernel's two-candidates-per-lane form, which costs 1.62-1.70x the one-candidate form's (tools/decode_r2.py),
// for two sequences.
template <typename PT, bool LM, bool GX, int R>
__global__ __launch_bounds__(64) void beam_search2_kernel(DecodeArgs a, int n_seq)
{
    // (R = 1: 117-129 VGPRs: beside two conv waves of 176-200 a SIMD has 112-160 left, so the wave fits next to the relu / match variants
    // and waits for a slot next to two residual ones; capping it at 96 -- amdgpu_waves_per_eu -- puts spills into scratch memory)
    constexpr int WM = R == 1 ? 6 : 12; // beams per sequence: 5 WM <= 32 R candidates
    constexpr int TS = 32;              // time steps per prepass tile: one lane of the half per step
    constexpr int KG = R == 1 ? 16 : 8; // keys per pass of the ranking loop
    constexpr int SEG = 32 + KG;        // doubles per key segment (<= 32 survivors + the padding of the last group)
    constexpr int LOG_TN = 9, TN = 1 << LOG_TN;
    static_assert(5 * WM <= 32 * R, "the candidates of a sequence fit its half");
    const int lane = threadIdx.x;
    const int h = lane >> 5;            // half = sequence slot of the wave
    const int hl = lane & 31;           // lane inside the half
    const int seq_raw = 2 * (int)blockIdx.x + h;
    const bool have = seq_raw < n_seq;
    const int seq = have ? seq_raw : n_seq - 1;          // (an odd last wave: the second half idles on valid addresses)
    const int T = have ? a.seq_len[seq] : 0;
    const PT* __restrict__ probs = (const PT*)a.probs;
    const int64_t row_a = a.seq_off[seq];
    const int64_t row_b = a.seq_off2 ? a.seq_off2[seq] : row_a;
    const int split = a.seq_off2 ? a.seq_split[seq] : 0;
    int4* __restrict__ childtab = a.childtab + a.node_off[seq];
    int* __restrict__ backptr = a.backptr + a.node_off[seq];
    const int W = a.W;
    const unsigned ctx_mask = LM ? ((a.k >= 16) ? 0xffffffffu : ((1u << (2 * a.k)) - 1u)) : 0u;
    const unsigned long long hmask = h ? 0xffffffff00000000ull : 0x00000000ffffffffull;

    __shared__ Beam st_[2][2][WM];
    __shared__ __attribute__((aligned(16))) double scr_[2][5 * WM];      // cpy_pnb | cpy_pb | mb_v | mP | mQ
    __shared__ __attribute__((aligned(16))) double keyC_[2][R * SEG];    // one key segment per candidate slot, in insertion order
    __shared__ int mb_q_[2][WM], d_sel_[2][WM], newslot_[2][WM];
    __shared__ unsigned claims_[2][WM];
    __shared__ double lp_[2][TS][5];
    __shared__ double praw_[LM ? 2 : 1][LM ? TS : 1][5];
    __shared__ double sent_[LM ? 2 : 1][LM ? TS : 1];
    __shared__ unsigned tab_[2][TN];
    __shared__ uint64_t gx_lds[GX ? 256 : 1];
    if constexpr (GX)
        for (int i = lane; i < 256; i += 64) gx_lds[i] = g_gm_exp_tab[i];
    const uint64_t* const gx_exp = gx_lds;

    double* const cpy_pnb = scr_[h];
    double* const cpy_pb = scr_[h] + WM;
    double* const mb_v = scr_[h] + 2 * WM;
    double* const mP = scr_[h] + 3 * WM;
    double* const mQ = scr_[h] + 4 * WM;
    double* const keyC = keyC_[h];
    int* const mb_q = mb_q_[h];
    int* const d_sel = d_sel_[h];
    int* const newslot = newslot_[h];
    unsigned* const claims = claims_[h];
    unsigned* const tab = tab_[h];
    double(*const lp)[5] = lp_[h];
    double(*const praw)[5] = praw_[LM ? h : 0];
    double* const sent = sent_[LM ? h : 0];

    for (int i = hl; i < TN; i += 32) tab[i] = i == 0 ? 0u : 0xffffffffu;    // (the empty labeling: id 0 in slot 0)
    if (hl == 0) {     // decode.py:128-132: the empty labeling with pr_blank = pr_total = log(1)
        Beam& b = st_[h][0][0];
        b.ptot = 0.0;
        b.pb = 0.0;
        b.pnb = -INFINITY;
        b.last = -1;
        b.len = 0;
        b.node = 0;
        b.hist = 0u;
        b.hprev = 0u;
        b.pad1 = 0;
        for (int c = 0; c < 4; c++) b.child[c] = 0;
        if (have) {
            childtab[0] = make_int4(0, 0, 0, 0);
            backptr[0] = 0;
        }
    }
    int nb = 1;              // beams currently kept (uniform within the half)
    int next_id = 1;         // next free trie node id of the half's sequence
    int cur = 0;             // (both halves flip together; a half that has ended stops writing)
    int fin = 0;             // which of the half's two record buffers holds its latest beam set (steps that change nothing but the scores do not flip)
    bool missed = false;     // sparse LM: see beam_search_kernel
    // the longer of the two sequences bounds the loop (wave-uniform)
    const int T_other = __shfl_xor(T, 32);
    const int Tmax = __builtin_amdgcn_readfirstlane(T > T_other ? T : T_other);
    wave_sync();

    for (int t0 = 0; t0 < Tmax; t0 += TS) {
        // ---- per-tile prepass: one lane of the half per time step (decode.py:165,168,193,195; :135-138)
        {
            const int t = t0 + hl;
            if (t < T) {
                const PT* __restrict__ prow = probs + ((t < split ? row_a : row_b) + t) * 5;
                double s4 = 0.0;
#pragma unroll 1
                for (int c = 0; c < 5; c++) {
                    const double pc = (double)prow[c];
                    lp[hl][c] = safe_log<GX>(pc);
                    if constexpr (LM) {
                        praw[hl][c] = pc;
                        if (c < 4) s4 = c == 0 ? pc : s4 + pc;
                    }
                }
                if constexpr (LM) {
                    const double sN = s4;
                    double ent = 0.0;
                    bool any = false;
#pragma unroll 1
                    for (int c = 0; c < 4; c++) {
                        const double pc = praw[hl][c];
                        double n;
                        if (sN == 0.0) n = pc;
                        else if constexpr (sizeof(PT) == 4) n = (double)((float)pc / (float)sN);
                        else n = pc / sN;
                        if (n > 0) {
                            double v = n * log_m<GX>(n);
                            ent = any ? ent + v : v;
                            any = true;
                        }
                    }
                    sent[hl] = any ? -ent : 0.0;
                }
            }
        }
        wave_sync();

        const int tend = (Tmax - t0) < TS ? (Tmax - t0) : TS;
        for (int tt = 0; tt < tend; tt++) {
            const bool live = t0 + tt < T;                  // this half's sequence still has a row (uniform within the half)
            const Beam* os = st_[h][cur];           // (no __restrict__: the scores-only step writes these records in place)
            Beam* __restrict__ ns = st_[h][cur ^ 1];
            const int ncand = live ? 5 * nb : 0;

            // ---------------- Phase A (beam_search_kernel, Phase A: same reads, same selects)
            bool valid[R], is_copy[R];
            int bi[R], kk[R], pj[R], dcopy[R], xch[R], chx[R];
            double c_ptot[R], c_pnb[R], c_pb[R], pnb_i[R], lpc[R];
            double2 pp[R];
            int2 ll[R];
            unsigned p_e[R];
            const double lp_blank = lp[tt][4];
            bool s_open = false;
            if constexpr (LM) s_open = sent[tt] > a.s_thr;
            const double ptot_last = os[nb - 1].ptot;
            const int myn = os[hl < nb ? hl : 0].node;
#pragma unroll
            for (int s = 0; s < R; s++) {
                const int q = s * 32 + hl;
                valid[s] = q < ncand;
                bi[s] = valid[s] ? q / 5 : 0;
                kk[s] = q - 5 * (q / 5);
                is_copy[s] = kk[s] == 0;
                pp[s] = *(const double2*)&os[bi[s]].ptot;
                pnb_i[s] = os[bi[s]].pnb;
                ll[s] = *(const int2*)&os[bi[s]].last;
                int4 c4 = *(const int4*)&os[bi[s]].child[0];
                asm("" : "+v"(c4.x), "+v"(c4.y), "+v"(c4.z), "+v"(c4.w));
                const int ci = (kk[s] - 1) & 3;
                const int c_lo = (ci & 1) ? c4.y : c4.x, c_hi = (ci & 1) ? c4.w : c4.z;
                chx[s] = (ci & 2) ? c_hi : c_lo;
            }
            if (hl < W) claims[hl] = 0u;
            const unsigned my_e = tab[myn & (TN - 1)];
#pragma unroll
            for (int s = 0; s < R; s++) {
                const int cc = is_copy[s] ? ll[s].x : kk[s] - 1;
                lpc[s] = lp[tt][cc < 0 ? 0 : cc];
                lpc[s] = cc < 0 ? -INFINITY : lpc[s];
                xch[s] = (valid[s] & !is_copy[s]) ? chx[s] : 0;
                p_e[s] = tab[xch[s] & (TN - 1)];
            }
            const bool self_bad = live & (hl < nb) & (my_e != ((((unsigned)myn >> LOG_TN) << 8) | (unsigned)hl));
            const bool tab_ok = (__ballot(self_bad) & hmask) == 0ull;
#pragma unroll
            for (int s = 0; s < R; s++) {
                const int last_i = ll[s].x;
                if constexpr (LM) {
                    const int len_i = ll[s].y;
                    const int cc = is_copy[s] ? last_i : kk[s] - 1;
                    const int need = is_copy[s] ? a.k + 1 : a.k;
                    if (s_open && valid[s] && cc >= 0 && len_i >= need) {
                        const unsigned hh = os[bi[s]].hist;
                        const unsigned ctx = (is_copy[s] ? (hh >> 2) : hh) & ctx_mask;
                        const bool gate = (a.lm_gate[ctx >> 5] >> (ctx & 31)) & 1u;
                        if (gate) {
                            const double r = a.lm_table[(size_t)ctx * 4 + cc];
                            double val;
                            if constexpr (sizeof(PT) == 4) {
                                const float f0 = (float)praw[tt][0], f1 = (float)praw[tt][1], f2 = (float)praw[tt][2], f3 = (float)praw[tt][3];
                                const float bp = ((f0 + f1) + f2) + f3;
                                const float sb = (float)praw[tt][cc] / bp;
                                val = ((r + (double)sb) / 2.0) * (double)bp;
                            } else {
                                const double bp = ((praw[tt][0] + praw[tt][1]) + praw[tt][2]) + praw[tt][3];
                                const double sb = praw[tt][cc] / bp;
                                val = ((r + sb) / 2.0) * bp;
                            }
                            lpc[s] = safe_log<GX>(val);
                        }
                    }
                }
                const double pnb_c = (last_i >= 0) ? pnb_i[s] + lpc[s] : -INFINITY;
                const double pb_c = pp[s].x + lp_blank;
                const double v = ((last_i == kk[s] - 1) ? pp[s].y : pp[s].x) + lpc[s];
                c_pnb[s] = is_copy[s] ? pnb_c : v;
                c_pb[s] = is_copy[s] ? pb_c : -INFINITY;
                c_ptot[s] = is_copy[s] ? 0.0 : v;
                dcopy[s] = is_copy[s] ? bi[s] : -1;
                if (valid[s] & is_copy[s]) {
                    cpy_pnb[bi[s]] = pnb_c;
                    cpy_pb[bi[s]] = pb_c;
                    mb_q[bi[s]] = -1;
                    newslot[bi[s]] = -1;
                }
                pj[s] = (tab_ok & (xch[s] != 0) & ((p_e[s] >> 8) == ((unsigned)xch[s] >> LOG_TN))) ? (int)(p_e[s] & 0xffu) : -1;
            }
            if (__any(!tab_ok)) {    // two kept beams of a half share a table entry (rare): that half compares against every beam
                for (int j = 0; j < WM; j++) {
                    const int nj = os[j < nb ? j : 0].node;
#pragma unroll
                    for (int s = 0; s < R; s++)
                        if (!tab_ok && j < nb && xch[s] != 0 && nj == xch[s]) pj[s] = j;
                }
            }
            bool any_merge = false;                         // in either half
#pragma unroll
            for (int s = 0; s < R; s++) any_merge |= __any(pj[s] >= 0) != 0;
            wave_sync();

            // ---------------- lae pass 1 / 2 (decode.py:174-175,199-201)
#pragma unroll
            for (int s = 0; s < R; s++) {
                const bool mext = pj[s] >= 0;
                const double cp = cpy_pnb[mext ? pj[s] : 0];
                const double x = mext ? cp : c_pb[s];
                const double y = mext ? c_ptot[s] : c_pnb[s];
                const double r = GX ? lae_gx(x, y, gx_exp) : lae(x, y);
                c_ptot[s] = is_copy[s] ? r : c_ptot[s];
                if (mext) {
                    mb_q[pj[s]] = s * 32 + hl;
                    mb_v[pj[s]] = c_ptot[s];
                    mQ[pj[s]] = r;
                }
            }
            if (any_merge) {
                wave_sync();
#pragma unroll
                for (int s = 0; s < R; s++) {
                    const int qe = mb_q[bi[s]];
                    const double mv = mb_v[bi[s]];
                    const bool m = valid[s] & is_copy[s] & (qe >= 0);
                    const double r = GX ? lae_gx(c_ptot[s], m ? mv : -INFINITY, gx_exp) : lae(c_ptot[s], m ? mv : -INFINITY);
                    if (m) mP[bi[s]] = r;
                }
                wave_sync();
#pragma unroll
                for (int s = 0; s < R; s++) {
                    const int q = s * 32 + hl;
                    const int j = is_copy[s] ? bi[s] : (pj[s] >= 0 ? pj[s] : 0);
                    const int qe = mb_q[j];
                    const double P = mP[j], Q = mQ[j], cb = cpy_pb[j];
                    const bool merged = valid[s] & (is_copy[s] ? qe >= 0 : pj[s] >= 0);
                    const int qother = is_copy[s] ? qe : 5 * j;
                    const bool mk = merged & (q < qother);
                    const bool mke = mk & !is_copy[s];
                    c_ptot[s] = mk ? P : c_ptot[s];
                    c_pnb[s] = mk ? Q : c_pnb[s];
                    c_pb[s] = mke ? cb : c_pb[s];
                    dcopy[s] = mke ? j : dcopy[s];
                    valid[s] = valid[s] & (!merged | mk);
                }
            }

            // ---------------- Phase D: rank by (pr_total desc, insertion order asc) among the candidates >= tau
            double key[R];
            bool surv[R];
            int lidx[R], scnt[R];
            int nvalid = 0;
            const double tau = (nb == W) ? ptot_last + lp_blank : -INFINITY;
#pragma unroll
            for (int s = 0; s < R; s++) {
                key[s] = valid[s] ? c_ptot[s] : __builtin_nan("");
                surv[s] = valid[s] && key[s] >= tau;
                const unsigned mv32 = (unsigned)((__ballot(valid[s]) & hmask) >> (32 * h));
                const unsigned ms32 = (unsigned)((__ballot(surv[s]) & hmask) >> (32 * h));
                nvalid += __popc(mv32);
                scnt[s] = __popc(ms32);
                lidx[s] = __popc(ms32 & ((1u << hl) - 1u));
                // the whole tail of the segment is padding (the other half may have more survivors and sets the loop's trip count)
                double* kseg = keyC + s * SEG;
                if (hl >= scnt[s]) kseg[hl] = -INFINITY;
                if (hl < KG) kseg[32 + hl] = -INFINITY;
            }
            wave_sync();
#pragma unroll
            for (int s = 0; s < R; s++)
                if (surv[s]) keyC[s * SEG + lidx[s]] = key[s];
            wave_sync();
            int smax[R], rank[R];
#pragma unroll
            for (int s = 0; s < R; s++) {
                const int scnt_o = __shfl_xor(scnt[s], 32);
                smax[s] = __builtin_amdgcn_readfirstlane(scnt[s] > scnt_o ? scnt[s] : scnt_o);
                rank[s] = 0;
            }
#pragma unroll
            for (int g = 0; g < R; g++) {
                const double* kseg = keyC + g * SEG;
                for (int j = 0; j < smax[g]; j += KG) {
                    double2 kq[KG / 2];
#pragma unroll
                    for (int u = 0; u < KG / 2; u++) kq[u] = *(const double2*)&kseg[j + 2 * u];
#pragma unroll
                    for (int u = 0; u < KG / 4; u++)
#pragma unroll
                        for (int s = 0; s < R; s++) rank[s] = count4_gt(rank[s], kq[2 * u].x, kq[2 * u].y, kq[2 * u + 1].x, kq[2 * u + 1].y, key[s]);
                }
            }
            const int nb_new = nvalid < W ? nvalid : W;

            // ---------------- the step that changes nothing but the scores (see beam_search_kernel): taken when BOTH halves qualify -- the two
            //                  sequences share the buffer flip -- where a half whose sequence has ended qualifies trivially
            if constexpr (R == 1) {
                const bool top = surv[0] && rank[0] < W;
                const unsigned long long m_top = __ballot(top) & hmask, m_same = __ballot(top && dcopy[0] >= 0 && rank[0] == dcopy[0]) & hmask;
                const bool ok_h = !live || (m_top == m_same && __popcll(m_top) == nb && nb_new == nb);
                if (__all(ok_h)) {
                    if (top) {
                        Beam* const here = st_[h][cur] + rank[0];
                        *(double2*)&here->ptot = make_double2(c_ptot[0], c_pb[0]);
                        here->pnb = c_pnb[0];
                    }
                    fin = live ? cur : fin;
                    wave_sync();
                    continue;
                }
            }

            // ---------------- Phase E: the kept candidates move to their new beam slot
            auto scatter = [&](bool on) {
#pragma unroll
                for (int s = 0; s < R; s++) {
                    if (on && surv[s] && rank[s] < W) {
                        const int r = rank[s];
                        *(double2*)&ns[r].ptot = make_double2(c_ptot[s], c_pb[s]);
                        ns[r].pnb = c_pnb[s];
                        d_sel[r] = (dcopy[s] & 0xff) | (bi[s] << 8) | (kk[s] << 16);
                        atomicAdd(&claims[r], 1u);
                        if (dcopy[s] >= 0) newslot[dcopy[s]] = r;
                    }
                }
            };
            scatter(true);
            wave_sync();
            const unsigned n_claims = claims[hl < nb_new ? hl : 0];
            int sel = d_sel[hl < nb_new ? hl : 0];
            int my_newslot = newslot[hl < nb ? hl : 0];
            unsigned my_tn2 = tab[myn & (TN - 1)];
            asm volatile("" : "+v"(sel), "+v"(my_newslot), "+v"(my_tn2));
            const bool tie_h = ((__ballot((hl < nb_new) & (n_claims != 1u)) & hmask) != 0ull);
            if (__any(tie_h)) {
                // equal keys claimed one slot: redo the count with the insertion-order rule -- in the half that has the tie
                int rank2[R];
#pragma unroll
                for (int s = 0; s < R; s++) rank2[s] = 0;
#pragma unroll
                for (int g = 0; g < R; g++) {
                    const double* kseg = keyC + g * SEG;
                    for (int j = 0; j < smax[g]; j++) {
                        const double kv = kseg[j];
#pragma unroll
                        for (int s = 0; s < R; s++) {
                            const bool before = g < s || (g == s && j < lidx[s]);      // insertion order = (slot segment, index in segment)
                            rank2[s] += (j < scnt[g] && ((kv > key[s]) || (kv == key[s] && before))) ? 1 : 0;
                        }
                    }
                }
#pragma unroll
                for (int s = 0; s < R; s++) rank[s] = tie_h ? rank2[s] : rank[s];
                if (tie_h && hl < nb) newslot[hl] = -1;
                wave_sync();
                scatter(tie_h);
                wave_sync();
                sel = d_sel[hl < nb_new ? hl : 0];
                my_newslot = newslot[hl < nb ? hl : 0];
            }

            // ---------------- Phase F: the new beam set: trie ids, labeling state
            {
                const bool act = live & (hl < nb_new);
                const int j = act ? (sel & 0xff) : 0;
                const int par = act ? (sel >> 8) & 0xff : 0;
                const int cl = act ? (sel >> 16) - 1 : 0;
                const bool is_ext = act && j == 0xff;
                const int src = is_ext ? par : j;
                const int4 meta = *(const int4*)&os[src].node;
                const int2 sl2 = *(const int2*)&os[src].last;
                const int4 chs = *(const int4*)&os[src].child[0];
                const int nid_old = os[src].child[cl & 3];
                const int ps = newslot[par];
                if (live & (hl < nb) & (my_newslot < 0) & ((my_tn2 >> 8) == ((unsigned)myn >> LOG_TN))) tab[myn & (TN - 1)] = 0xffffffffu;
                const bool fresh = is_ext && nid_old == 0;
                const bool reload = is_ext && nid_old != 0;
                const unsigned fm32 = (unsigned)((__ballot(fresh) & hmask) >> (32 * h));
                const int my_node = fresh ? next_id + __popc(fm32 & ((1u << hl) - 1u)) : nid_old;
                if (fresh) {
                    backptr[my_node] = (meta.x << 2) | cl;
                    ((int*)&childtab[meta.x])[cl] = my_node;
                    childtab[my_node] = make_int4(0, 0, 0, 0);
                }
                next_id += __popc(fm32);
                int4 ch = make_int4(0, 0, 0, 0);
                if (__any(reload)) {
                    __builtin_amdgcn_s_waitcnt(0);
                    if (reload) {
                        const int* cp = (const int*)&childtab[my_node];
                        ch.x = __hip_atomic_load(cp + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ch.y = __hip_atomic_load(cp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ch.z = __hip_atomic_load(cp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ch.w = __hip_atomic_load(cp + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(ch.x), "+v"(ch.y), "+v"(ch.z), "+v"(ch.w));
                }
                if (act) {
                    const int new_node = is_ext ? my_node : meta.x;
                    *(int2*)&ns[hl].last = make_int2(is_ext ? cl : sl2.x, is_ext ? sl2.y + 1 : sl2.y);
                    const unsigned h_new = ((unsigned)meta.y << 2) | (unsigned)cl;
                    if constexpr (LM) {
                        if (a.lm_missing && is_ext && sl2.y + 1 >= a.k && t0 + tt + 1 < T) {
                            const unsigned cx = h_new & ctx_mask;
                            missed |= ((a.lm_missing[cx >> 5] >> (cx & 31)) & 1u) != 0u;
                        }
                    }
                    *(int4*)&ns[hl].node = make_int4(new_node, is_ext ? (int)h_new : meta.y, meta.z, 0);
                    *(int4*)&ns[hl].child[0] = make_int4(is_ext ? ch.x : chs.x, is_ext ? ch.y : chs.y, is_ext ? ch.z : chs.z, is_ext ? ch.w : chs.w);
                    tab[new_node & (TN - 1)] = (((unsigned)new_node >> LOG_TN) << 8) | (unsigned)hl;
                }
                if (fresh && ps >= 0) ns[ps].child[cl] = my_node;
            }
            nb = live ? nb_new : nb;
            cur ^= 1;
            fin = live ? cur : fin;
            wave_sync();
        }
    }

    // ---------------- traceback of each half's best labeling (decode.py:207-210)
    int hl_end = hl;
    asm volatile("" : "+v"(hl_end));
    const bool any_missed = LM && (__ballot(missed) & hmask) != 0ull;
    if (hl_end == 0 && have) {
        __builtin_amdgcn_s_waitcnt(0);
        const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        const DecodeArgs* ap = (const DecodeArgs*)ka;
        const Beam& fs = st_[h][fin][0];                // (the buffer the half's last live step left its records in)
        int n = fs.node;
        const int len = fs.len;
        uint8_t* out = ap->labels + ap->label_off[seq];
        for (int p = len - 1; p >= 0; p--) {
            const int bp = __hip_atomic_load(&backptr[n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            out[p] = (uint8_t)(bp & 3);
            n = bp >> 2;
        }
        ap->label_len[seq] = any_missed ? -1 : len;
        if (ap->best_score) ap->best_score[seq] = fs.ptot;
    }
}

// Launch shape.  W <= 12: one wave per sequence.  Wider beams have two forms: several waves per sequence (two for W <= 25,
// four for W <= 51; one candidate per lane) -- the shortest time step, for launches that leave SIMDs idle (global decode of
// a batch of reads) -- and fewer waves with two candidates per lane, which issues fewer instructions per sequence and keeps
// more sequences resident: the form for launches of thousands of sequences (chunk decode of a group of batches).
// (rd_set_decode_form pins one form, for tests and measurements.)
constexpr int kMaxW = Cfg<1, 4>::WM;
static_assert(Cfg<2, 2>::WM == kMaxW, "both forms cover the same widths");
constexpr int kMaxW2 = Cfg<2, 4>::WM;   // 64: widths 52 ... 64 on four waves with two candidates per lane
constexpr int kMaxW3 = Cfg<2, 5>::WM;   // 128: widths 65 ... 128 on five waves with two candidates per lane, the beam set in two halves (round 6)
constexpr int kMaxW4 = Cfg<2, 10>::WM;  // 256: widths 129 ... 256 on ten waves, the beam set in four parts (one workgroup per CU: 81 KB of LDS)
static_assert(kMaxW4 == RD_LANE_MAX_W && kMaxW2 == RD_HASHED_MAX_W, "common.h's figures are this file's");

template <typename PT, int R, int NW, bool GX>
int launch_g(hipStream_t st, const DecodeArgs& a, int n_seq, bool lm)
{
    if (lm && a.hashed)
        hipLaunchKernelGGL((beam_search_kernel<PT, R, NW, true, true, GX>), dim3(n_seq), dim3(64 * NW), 0, st, a);
    else if (lm)
        hipLaunchKernelGGL((beam_search_kernel<PT, R, NW, true, false, GX>), dim3(n_seq), dim3(64 * NW), 0, st, a);
    else
        hipLaunchKernelGGL((beam_search_kernel<PT, R, NW, false, false, GX>), dim3(n_seq), dim3(64 * NW), 0, st, a);
    RD_HIP(hipGetLastError());
    return RD_OK;
}

template <typename PT, int R, int NW>
int launch_r(hipStream_t st, const DecodeArgs& a, int n_seq, bool lm)
{
    return a.glibc_math ? launch_g<PT, R, NW, true>(st, a, n_seq, lm) : launch_g<PT, R, NW, false>(st, a, n_seq, lm);
}

// work-queue form (beam_search_queue_kernel): `slots` workgroups take n_seq sequences; no hashed contexts (the pipeline's groups have none
// beyond the partition's capacity: configs[4]'s leg closes at the limit, as before)
template <typename PT, int R, int NW, bool GX>
int launch_qg(hipStream_t st, const DecodeArgs& a, int n_seq, bool lm, int slots, int* counter)
{
    if (lm) hipLaunchKernelGGL((beam_search_queue_kernel<PT, R, NW, true, false, GX>), dim3(slots), dim3(64 * NW), 0, st, a, n_seq, counter);
    else hipLaunchKernelGGL((beam_search_queue_kernel<PT, R, NW, false, false, GX>), dim3(slots), dim3(64 * NW), 0, st, a, n_seq, counter);
    RD_HIP(hipGetLastError());
    return RD_OK;
}
template <typename PT>
int launch_queue_pt(hipStream_t st, const DecodeArgs& a, int n_seq, bool lm, int wave_slots, int* counter)
{
    // one workgroup = 1 / 2 / 4 waves for W <= 12 / 25 / 51 (the shapes part_seq_limit counts with); round 6: 4 waves with two candidates per
    // lane up to 64, 5 waves up to 128 (a global-mode stream at W = 100 whose groups could not go through the queue fell to 2.9 M samples/s
    // on alternating read lengths: the groups closed at the partition's sequence limit, covered or not)
    const int nw = a.W <= Cfg<1, 1>::WM ? 1 : a.W <= Cfg<1, 2>::WM ? 2 : a.W <= kMaxW2 ? 4 : a.W <= kMaxW3 ? 5 : 10;
    const int slots = std::max(1, std::min(n_seq, wave_slots / nw));
    if (a.W > kMaxW3) return a.glibc_math ? launch_qg<PT, 2, 10, true>(st, a, n_seq, lm, slots, counter) : launch_qg<PT, 2, 10, false>(st, a, n_seq, lm, slots, counter);
    if (a.W > kMaxW2) return a.glibc_math ? launch_qg<PT, 2, 5, true>(st, a, n_seq, lm, slots, counter) : launch_qg<PT, 2, 5, false>(st, a, n_seq, lm, slots, counter);
    if (a.W > kMaxW) return a.glibc_math ? launch_qg<PT, 2, 4, true>(st, a, n_seq, lm, slots, counter) : launch_qg<PT, 2, 4, false>(st, a, n_seq, lm, slots, counter);
    if (nw == 1) return a.glibc_math ? launch_qg<PT, 1, 1, true>(st, a, n_seq, lm, slots, counter) : launch_qg<PT, 1, 1, false>(st, a, n_seq, lm, slots, counter);
    if (nw == 2) return a.glibc_math ? launch_qg<PT, 1, 2, true>(st, a, n_seq, lm, slots, counter) : launch_qg<PT, 1, 2, false>(st, a, n_seq, lm, slots, counter);
    return a.glibc_math ? launch_qg<PT, 1, 4, true>(st, a, n_seq, lm, slots, counter) : launch_qg<PT, 1, 4, false>(st, a, n_seq, lm, slots, counter);
}

template <typename PT, int R>
int launch_two(hipStream_t st, const DecodeArgs& a, int n_seq, bool lm)
{
    const dim3 grid((unsigned)((n_seq + 1) / 2));
    if (lm) {
        if (a.glibc_math) hipLaunchKernelGGL((beam_search2_kernel<PT, true, true, R>), grid, dim3(64), 0, st, a, n_seq);
        else hipLaunchKernelGGL((beam_search2_kernel<PT, true, false, R>), grid, dim3(64), 0, st, a, n_seq);
    } else {
        if (a.glibc_math) hipLaunchKernelGGL((beam_search2_kernel<PT, false, true, R>), grid, dim3(64), 0, st, a, n_seq);
        else hipLaunchKernelGGL((beam_search2_kernel<PT, false, false, R>), grid, dim3(64), 0, st, a, n_seq);
    }
    RD_HIP(hipGetLastError());
    return RD_OK;
}

// 65 ... 128 beams: no hashed contexts (refused where the arguments are checked: RD_REQUIRE_WIDTH_LM)
template <typename PT, int NW>
int launch_128(hipStream_t st, const DecodeArgs& a, int n_seq, bool lm)
{
    if (lm) {
        if (a.glibc_math) hipLaunchKernelGGL((beam_search_kernel<PT, 2, NW, true, false, true>), dim3(n_seq), dim3(64 * NW), 0, st, a);
        else hipLaunchKernelGGL((beam_search_kernel<PT, 2, NW, true, false, false>), dim3(n_seq), dim3(64 * NW), 0, st, a);
    } else {
        if (a.glibc_math) hipLaunchKernelGGL((beam_search_kernel<PT, 2, NW, false, false, true>), dim3(n_seq), dim3(64 * NW), 0, st, a);
        else hipLaunchKernelGGL((beam_search_kernel<PT, 2, NW, false, false, false>), dim3(n_seq), dim3(64 * NW), 0, st, a);
    }
    RD_HIP(hipGetLastError());
    return RD_OK;
}

template <typename PT>
int launch_pt(hipStream_t st, const DecodeArgs& a, int n_seq, bool lm, int n_simd, int form)
{
    if (a.W > kMaxW2) {
        RD_REQUIRE(!(lm && a.hashed), "beam widths above %d do not combine with hashed long contexts (rd_load_lm_hashed)", kMaxW2);
        return a.W > kMaxW3 ? launch_128<PT, 10>(st, a, n_seq, lm) : launch_128<PT, 5>(st, a, n_seq, lm);
    }
    // W <= 6: two sequences per wave.  Measured (tools/decode_bench.py, 1024-row windows, W = 6): 4096 windows 1.23 -> 1.80 G time
    // steps/s (glibc arithmetic 1.10 -> 1.55 G, soft rows 0.94 -> 1.32 G); 512 windows -- lone waves, the latency case -- 1.58 vs
    // 1.60 ms per launch: a step of the two-sequence wave is as short as the one-sequence wave's, so there is no case for the
    // latter (rd_set_decode_form 4 keeps it reachable for tests and A/B runs; 3 = the default's choice, spelled out)
    if (a.W <= 6 && !(lm && a.hashed) && form != 4 && n_seq >= 2) return launch_two<PT, 1>(st, a, n_seq, lm);
    // 7 <= W <= 12: two sequences per wave with two candidates per lane of the half exists (form 3) and is NOT the default's choice.
    // Measured (round 4; tools/decode_r2.py, tools/bench_ab2.sh; profiles/r04_beam_search_two_r2.txt): 4096 windows x 1024 rows at
    // W = 10 alone on the chip 1005 M time steps/s against 1190 M one per wave (glibc arithmetic 859 against 1079 M); the headline
    // loop 26.2 against 26.6 M samples/s.  Its instruction stream is 1.62-1.70x the one-candidate form's for two sequences (0.82x
    // per sequence), but it halves the waves that hide each other's LDS round trips, and at 131-145 VGPRs the wave no longer fits
    // beside two conv waves of 206 on a SIMD (the one-sequence wave: 82-103).
    if (a.W <= 12 && !(lm && a.hashed) && n_seq >= 2 && form == 3) return launch_two<PT, 2>(st, a, n_seq, lm);
    (void)n_simd;
    if (a.W <= Cfg<1, 1>::WM) return form == 2 ? launch_r<PT, 2, 1>(st, a, n_seq, lm) : launch_r<PT, 1, 1>(st, a, n_seq, lm);   // (form 2 here: measurements only)
    // wide form while every wave still gets a SIMD of its own
    const bool mid = a.W <= Cfg<1, 2>::WM;
    const bool wide = form == 1 || (form == 0 && (long long)n_seq * (mid ? 2 : 4) <= n_simd);
    if (mid) return wide ? launch_r<PT, 1, 2>(st, a, n_seq, lm) : launch_r<PT, 2, 1>(st, a, n_seq, lm);
    if (a.W > kMaxW) return launch_r<PT, 2, 4>(st, a, n_seq, lm);      // 52 ... 64
    return wide ? launch_r<PT, 1, 4>(st, a, n_seq, lm) : launch_r<PT, 2, 2>(st, a, n_seq, lm);
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

extern "C" int rd_decode_max_width(void) { return RD_WIDE_MAX_W; }
extern "C" int rd_decode_lane_width(void) { return kMaxW4; }

// Per-context gate bits: bit = (entropy(lm[ctx]) < r_threshold)   decode.py:85-93.
// The entropies were computed once at rd_load_lm (glibc log, like the reference's math.log) and live in HBM.
static int ensure_lm_gate(rd_ctx* ctx, double r_thr, hipStream_t st)
{
    LM& lm = ctx->lm;
    if (lm.gate_valid && lm.gate_r_thr == r_thr) return RD_OK;
    const size_t n = (size_t)1 << (2 * lm.table_order);
    const size_t words = (n + 31) / 32;
    if (lm.gate_storage.reserve(words * 4)) return RD_ERR_NOMEM;
    lm.gate_bits = (uint32_t*)lm.gate_storage.p;
    const int threads = 256;
    const int blocks = (int)((words + threads - 1) / threads);
    hipLaunchKernelGGL(lm_gate_kernel, dim3(blocks), dim3(threads), 0, st, lm.d_entropy, n, r_thr, lm.gate_bits);
    RD_HIP(hipGetLastError());
    lm.gate_valid = true;
    lm.gate_r_thr = r_thr;
    return RD_OK;
}

int rd_decode_dev(rd_ctx* ctx, const void* d_probs, int ptype, const int64_t* d_seq_off, const int32_t* d_seq_len,
                  const int64_t* d_node_off, const int64_t* d_label_off, int n_seq, int64_t total_nodes, int W, int use_lm,
                  double s_thr, double r_thr, uint8_t* d_labels, int32_t* d_label_len, double* d_best_score, hipStream_t stream,
                  const int64_t* d_seq_off2, const int32_t* d_seq_split, int n_cu_avail, int queue_wave_slots)
{
    const int n_simd = 4 * (n_cu_avail > 0 ? n_cu_avail : ctx->n_cu);
    hipStream_t st = stream ? stream : ctx->stream;
    RD_REQUIRE(W >= 1 && W <= RD_WIDE_MAX_W, "beam_width %d out of range [1,%d]", W, RD_WIDE_MAX_W);
    if (n_seq == 0) return RD_OK;
    if (use_lm) {
        if (!ctx->lm.loaded) {
            rd_set_error("decode with use_lm=1 but no LM table loaded (rd_load_lm)");
            return RD_ERR_STATE;
        }
        int rc = ensure_lm_gate(ctx, r_thr, st);
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
    a.glibc_math = ctx->decode_math;
    a.lm_table = use_lm ? ctx->lm.table : nullptr;
    a.lm_gate = use_lm ? ctx->lm.gate_bits : nullptr;
    a.lm_missing = (use_lm && ctx->lm.sparse && !ctx->lm.hashed) ? ctx->lm.d_missing : nullptr;
    a.k = use_lm ? ctx->lm.k : 0;
    a.hashed = use_lm ? ctx->lm.hashed : 0;
    a.tmask = use_lm ? (unsigned)(((size_t)1 << (2 * ctx->lm.table_order)) - 1) : 0u;
    a.bk = 1u;
    for (int i = 0; use_lm && i < ctx->lm.k; i++) a.bk *= kHashB;
    a.s_thr = s_thr;
    a.childtab = ctx->ws_nodes_child.as<int4>();
    a.backptr = ctx->ws_nodes_back.as<int>();
    a.labels = d_labels;
    a.label_len = d_label_len;
    a.best_score = d_best_score;
    KernelTimer& tm = ctx->timer_decode;
    if (tm.enabled && tm.used < tm.starts.size()) RD_HIP(hipEventRecord(tm.starts[tm.used], st));
    int rc;
    if (ctx->decode_form == 5 && queue_wave_slots == 0) queue_wave_slots = 16;   // rd_set_decode_form 5 (tests): every launch through the work queue, few slots
    if (queue_wave_slots > 0 && W <= kMaxW4 && !(use_lm && a.hashed)) {
        // more sequences than the CUs of this stream keep resident: resident workgroups + a work queue (beam_search_queue_kernel)
        if (ctx->ws_queue.reserve(256)) return RD_ERR_NOMEM;
        RD_HIP(hipMemsetAsync(ctx->ws_queue.p, 0, 4, st));
        int* counter = ctx->ws_queue.as<int>();
        rc = ptype == 1 ? launch_queue_pt<double>(st, a, n_seq, use_lm != 0, queue_wave_slots, counter)
             : ptype == 2 ? launch_queue_pt<_Float16>(st, a, n_seq, use_lm != 0, queue_wave_slots, counter)
                          : launch_queue_pt<float>(st, a, n_seq, use_lm != 0, queue_wave_slots, counter);
    } else
    rc = W > kMaxW4 ? rd_decode_wide_launch(ctx, st, &a, ptype, n_seq, total_nodes, use_lm != 0)   // (decode_wide.hip: any wider beam)
             : ptype == 1 ? launch_pt<double>(st, a, n_seq, use_lm != 0, n_simd, ctx->decode_form)
             : ptype == 2 ? launch_pt<_Float16>(st, a, n_seq, use_lm != 0, n_simd, ctx->decode_form) : launch_pt<float>(st, a, n_seq, use_lm != 0, n_simd, ctx->decode_form);
    if (rc) return rc;
    if (tm.enabled && tm.used < tm.starts.size()) {
        RD_HIP(hipEventRecord(tm.stops[tm.used], st));
        tm.used++;
    }
    return RD_OK;
}
