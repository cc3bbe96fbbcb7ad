// plan.hip -- host-side geometry of the reads-level paths: see plan.h.
#include "plan.h"

namespace rdi {

int64_t assembled_rows(int nW, int T, int pad, int step)
{
    int64_t N = 0;
    for (int i = 0; i < nW; i++) {
        int rows = (i == nW - 1) ? T - pad : T;
        int64_t end = (int64_t)i * step + rows;
        if (rows > 0 && end > N) N = end;
    }
    return N;
}

int assembled_is_f64(int nW, int T, int pad, int step)
{
    // some time step is covered twice <=> a window i >= 1 with rows starts inside window i-1
    if (step >= T) return 0;
    if (nW >= 3) return 1;
    if (nW == 2 && T - pad > 0) return 1;
    return 0;
}


void add_segment(std::vector<TileDesc>& list, int64_t& rows, int64_t seg_row, int64_t src_row, int len, int in_len,
                 int64_t alt_row = 0, int alt_in = INT32_MAX, int alt_res = INT32_MAX)
{
    for (int t0 = 0; t0 < len; t0 += 32) {   // 32-row sub-tiles; four of them (of any segments) make a workgroup tile
        TileDesc td;
        td.seg_row = seg_row;
        td.src_row = src_row;
        td.alt_row = alt_row;
        td.t0 = t0;
        td.seg_len = len;
        td.in_len = in_len;
        td.alt_in = alt_in;
        td.alt_res = alt_res;
        td.pad_ = 0;
        list.push_back(td);
    }
    rows += len;
}

// Rows of a head that differ from the stream, per tensor: the signal has none; a k=3 conv of dilation d adds 2d.
struct LayerHalo {
    int h_in, h_res, h_out;
};
void layer_halos(const Model& m, LayerHalo* lh)
{
    int H = 0;  // halo of the block input
    for (int b = 0; b < m.nblocks; b++) {
        const int d = m.dil[b];
        lh[2 * b] = {H, H, H + 2 * d};               // first conv (block 0: from the raw signal)
        lh[2 * b + 1] = {H + 2 * d, H, H + 4 * d};   // second conv, residual = block input
        H += 4 * d;
    }
    lh[2 * m.nblocks] = {H, H, H};                   // dense head
}

// chunk mode: one stream per read + one head per window i >= 1; per-layer head lengths
int plan_reads_chunk(const Model& m, const int64_t* read_off, int n_reads, int chunk, int step, int halo, ReadsPlan& P)
{
    LayerHalo lh[RD_MAX_LAYERS];
    layer_halos(m, lh);
    P.n_layers = 2 * m.nblocks + 1;
    P.per_layer = true;
    int64_t row = 0;
    P.read_win_off.assign(1, 0);
    // A layer's tile list holds the streams of ALL reads first, then the heads: workgroups are dispatched in list order as
    // slots free up, so the full 128-row stream tiles fill the rounds and the short head tiles (few rows, waves without rows
    // skip their MFMAs) make up the last, partial round -- longest-processing-time-first.  (Tiles of one launch are
    // independent: a head reads its later rows from the layer's INPUT tensor.)
    std::vector<TileDesc> heads[RD_MAX_LAYERS];
    for (int r = 0; r < n_reads; r++) {
        const int64_t N = read_off[r + 1] - read_off[r];
        RD_REQUIRE(N >= 1, "read %d is empty", r);
        RD_REQUIRE(N < INT32_MAX, "read %d too long", r);
        const WindowGeom g = window_geom(N, chunk, step);
        const int64_t stream_row = row;
        P.read_row.push_back(stream_row);
        for (int li = 0; li < P.n_layers; li++) add_segment(P.tiles[li], P.rows[li], stream_row, read_off[r], (int)N, (int)N);
        row += N;
        for (int i = 0; i < g.nW; i++) {
            const int valid = (i < g.nW - 1) ? chunk : chunk - g.pad;
            const int h = (i == 0) ? 0 : (halo < valid ? halo : valid);
            int64_t o1 = stream_row + (int64_t)i * step;
            if (h > 0) {
                o1 = row;
                const int64_t alt = stream_row + (int64_t)i * step;
                for (int li = 0; li < P.n_layers; li++) {
                    const int len = lh[li].h_out < valid ? lh[li].h_out : valid;   // rows of this head the layer must produce
                    if (len > 0)
                        add_segment(heads[li], P.rows[li], row, read_off[r] + (int64_t)i * step, len, valid, alt, lh[li].h_in, lh[li].h_res);
                }
                row += h;
            }
            P.off1.push_back(o1);
            P.off2.push_back(stream_row + (int64_t)i * step);
            P.split.push_back(h);
            P.valid.push_back(valid);
        }
        P.n_windows += g.nW;
        P.read_win_off.push_back(P.n_windows);
    }
    for (int li = 0; li < P.n_layers; li++) P.tiles[li].insert(P.tiles[li].end(), heads[li].begin(), heads[li].end());
    P.total_rows = row;
    return RD_OK;
}

// global mode: one stream per read when the geometry allows it, else per-window segments in a uniform row layout
int plan_reads_global(const Model& m, const int64_t* read_off, int n_reads, int chunk, int step, int halo, ReadsPlan& P, bool* streamed)
{
    const bool st = step <= chunk - halo;
    *streamed = st;
    P.n_layers = 2 * m.nblocks + 1;
    P.per_layer = false;
    int64_t row = 0;
    P.read_win_off.assign(1, 0);
    for (int r = 0; r < n_reads; r++) {
        const int64_t N = read_off[r + 1] - read_off[r];
        RD_REQUIRE(N >= 1, "read %d is empty", r);
        RD_REQUIRE(N < INT32_MAX, "read %d too long", r);
        const WindowGeom g = window_geom(N, chunk, step);
        P.read_row.push_back(row);
        if (st) {
            add_segment(P.tiles[0], P.rows[0], row, read_off[r], (int)N, (int)N);
            row += N;
        } else {
            for (int i = 0; i < g.nW; i++) {
                const int valid = (i < g.nW - 1) ? chunk : chunk - g.pad;
                if (valid > 0) add_segment(P.tiles[0], P.rows[0], row + (int64_t)i * chunk, read_off[r] + (int64_t)i * step, valid, valid);
            }
            row += (int64_t)g.nW * chunk;
        }
        P.valid.push_back(g.pad);   // per read: the pad of its last window
        P.n_windows += g.nW;
        P.read_win_off.push_back(P.n_windows);
    }
    P.total_rows = row;
    return RD_OK;
}

size_t plan_pad_tiles(ReadsPlan& P)
{
    size_t total = 0;
    for (int li = 0; li < P.n_layers; li++)
        if (P.per_layer || li == 0) {
            std::vector<TileDesc>& v = P.tiles[li];
            while (v.size() % 8) {
                TileDesc e = {};
                e.alt_in = e.alt_res = INT32_MAX;
                v.push_back(e);
            }
            total += v.size();
        }
    return total;
}

int pinned_reserve(void** p, size_t* cap, size_t bytes)
{
    if (bytes <= *cap) return RD_OK;
    if (*p) (void)hipHostFree(*p);
    *p = nullptr;
    *cap = 0;
    size_t want = align_up(bytes + bytes / 8, 1 << 16);
    RD_HIP(hipHostMalloc(p, want, hipHostMallocDefault));
    *cap = want;
    return RD_OK;
}

}  // namespace rdi
