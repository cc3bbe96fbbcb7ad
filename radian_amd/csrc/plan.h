// plan.h -- host-side geometry of the reads-level paths (internal; shared by api.hip and pipe_reads.hip):
// windows of a read (preprocess.py:4-22), assembled length / dtype of a read (matrix_assembly.py:6-53), and the
// segment / tile-descriptor plans of the streamed forward.
#pragma once
#include "common.h"

namespace rdi {

int64_t assembled_rows(int nW, int T, int pad, int step);
int assembled_is_f64(int nW, int T, int pad, int step);

inline int count_windows(int64_t N, int chunk, int step) { return (N < chunk ? 0 : (int)((N - chunk) / step) + 1) + 1; }

struct WindowGeom {
    int nW, pad;
};
inline WindowGeom window_geom(int64_t N, int chunk, int step)
{
    WindowGeom g;
    g.nW = count_windows(N, chunk, step);
    const int64_t last_start = (int64_t)(g.nW - 1) * step;
    g.pad = (int)(chunk - (N - last_start));   // >= 1 always (preprocess.py:17-19)
    return g;
}

struct ReadsPlan {
    int n_layers = 0;                       // 2 * nblocks + 1
    bool per_layer = false;                 // false: tiles[0] serves every layer
    std::vector<TileDesc> tiles[RD_MAX_LAYERS];
    int64_t rows[RD_MAX_LAYERS] = {0};      // time steps evaluated per layer
    // per decoded sequence (chunk mode: window; global mode: read)
    std::vector<int64_t> off1, off2;
    std::vector<int32_t> split, valid;
    std::vector<int32_t> read_win_off;   // n_reads + 1
    std::vector<int64_t> read_row;       // first stream row of each read
    int64_t total_rows = 0;
    int n_windows = 0;
};

// chunk mode: one stream per read + one head per window i >= 1; per-layer head lengths
int plan_reads_chunk(const Model& m, const int64_t* read_off, int n_reads, int chunk, int step, int halo, ReadsPlan& P);
// global mode: one stream per read when the geometry allows it (*streamed), else per-window segments
int plan_reads_global(const Model& m, const int64_t* read_off, int n_reads, int chunk, int step, int halo, ReadsPlan& P, bool* streamed);
// pads every tile list the plan uses to whole workgroup tiles (eight sub-tiles: the bf16x3 kernel's tile); -> descriptors in total
size_t plan_pad_tiles(ReadsPlan& P);

// grow-only pinned host buffer
int pinned_reserve(void** p, size_t* cap, size_t bytes);

}  // namespace rdi
