// fast5.hip -- host code: the raw signals of a fast5 file, radian/basecall.py:7,70-76 (`get_fast5_file(path).get_reads()`,
// `read.get_raw_data()`, `read.read_id`), read in BATCHES straight out of the file's mapping into one int16 block + offsets.
//
// Why native: through libhdf5 a read costs ~63 us (group walk, dataset open, H5Dread) plus the per-read Python / ctypes / interpreter-lock
// hand-off -- 11 M samples/s per core, so eight GPUs at 27 M samples/s each need ~20 cores for parsing alone.  A fast5 written with
// libhdf5's default settings (what ont_fast5_api / h5py's `libver="earliest"` produce, and what the reference's sample
// radian/data/reads.fast5 is) uses the CLASSIC HDF5 layout: superblock version 0/1, version-1 object headers with continuation blocks,
// symbol-table groups (v1 B-tree of SNOD leaves + a local heap of names), "new-style" groups whose links sit compactly in a v1 header,
// contiguous / compact / chunked (v1 chunk B-tree) datasets, the chunks raw or run through HDF5's built-in filters -- deflate (the gzip
// level-1 signals of pre-VBZ MinKNOW files), byte shuffle, Fletcher-32.  That layout is a handful of pointer chases per read; this file walks it
// over a read-only mapping: one call resolves and copies a block of reads (~2 us per 4096-sample read, ~25 us when it has to inflate),
// with the interpreter lock released.
//
// ANYTHING else -- superblock >= 2, version-2 object headers, dense (fractal-heap) groups, any other filter (VBZ = 32020, szip, n-bit,
// scale-offset), a chunk that does not inflate to exactly one chunk or fails its checksum, a signal that is not little-endian int16, or any offset that points outside the file -- returns RD_ERR_FORMAT and decides
// NOTHING: the caller (radian_amd/fast5.py) reads that file through libhdf5 as before, whose errors are then the verdict.
//
// Read order = ont_fast5_api's = HDF5's name order: multi-read files iterate the root's `read_<id>` groups by name (id = the name without the
// prefix; signal at Raw/Signal), single-read files the groups of /Raw/Reads (id = the group's `read_id` attribute, else its name).
//
// Every access to the mapping is bounds-checked (tests/asan_fast5.cpp feeds mutated and truncated files in exact-size heap buffers).
// No GPU is touched; the file is part of libradian_hip.so so that the host side stays one ctypes binding.
#include "common.h"
#include "../../include/radian_hip.h"

#include <algorithm>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>
#include <zlib.h>

namespace {

struct NoVerdict {          // thrown inside, turned into RD_ERR_FORMAT at the C boundary
    const char* why;
};

constexpr uint64_t kUndef = 0xFFFFFFFFFFFFFFFFull;
constexpr uint8_t kSig[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};

struct Image {
    const uint8_t* p = nullptr;
    uint64_t n = 0;
    uint64_t base = 0;      // superblock base address: every file address is relative to it

    void need(uint64_t off, uint64_t len) const
    {
        if (off > n || len > n - off) throw NoVerdict{"an address points outside the file"};
    }
    uint8_t u8(uint64_t o) const
    {
        need(o, 1);
        return p[o];
    }
    uint16_t u16(uint64_t o) const
    {
        need(o, 2);
        uint16_t v;
        memcpy(&v, p + o, 2);
        return v;
    }
    uint32_t u32(uint64_t o) const
    {
        need(o, 4);
        uint32_t v;
        memcpy(&v, p + o, 4);
        return v;
    }
    uint64_t u64(uint64_t o) const
    {
        need(o, 8);
        uint64_t v;
        memcpy(&v, p + o, 8);
        return v;
    }
    uint64_t addr(uint64_t o) const     // a file address stored at o -> offset into the mapping
    {
        const uint64_t a = u64(o);
        if (a == kUndef) return kUndef;
        if (a > n || base > n - a) throw NoVerdict{"an address points outside the file"};
        return a + base;
    }
    bool tag(uint64_t o, const char* t) const
    {
        need(o, 4);
        return memcmp(p + o, t, 4) == 0;
    }
};

struct Msg {
    uint16_t type;
    uint8_t flags;      // bit 1: the message is SHARED (its body is a reference to another object header / the shared-message heap)
    uint64_t data;
    uint32_t size;
};

// version-1 object header -> its messages, following continuation blocks
void header_messages(const Image& im, uint64_t at, std::vector<Msg>& out)
{
    out.clear();
    if (im.u8(at) != 1) {
        if (im.tag(at, "OHDR")) throw NoVerdict{"version-2 object header"};
        throw NoVerdict{"not an object header"};
    }
    const unsigned nmsg = im.u16(at + 2);
    struct Block {
        uint64_t p, n;
    };
    std::vector<Block> blocks{{at + 16, im.u32(at + 8)}};
    size_t bi = 0;
    while (bi < blocks.size() && out.size() < nmsg) {
        if (blocks.size() > 256) throw NoVerdict{"object header with too many continuation blocks"};
        uint64_t p = blocks[bi].p;
        const uint64_t n = blocks[bi].n;
        bi++;
        im.need(p, n);
        const uint64_t end = p + n;
        while (p + 8 <= end && out.size() < nmsg) {
            const uint16_t mtype = im.u16(p), msize = im.u16(p + 2);
            const uint64_t data = p + 8;
            if (msize > end - data) throw NoVerdict{"header message runs past its block"};
            if (mtype == 0x0010) {   // continuation: address, length
                if (msize < 16) throw NoVerdict{"short continuation message"};
                const uint64_t ca = im.addr(data);
                if (ca == kUndef) throw NoVerdict{"undefined continuation address"};
                blocks.push_back({ca, im.u64(data + 8)});
            }
            out.push_back({mtype, im.u8(p + 4), data, msize});
            p = data + msize;
        }
    }
}

const Msg* find_msg(const std::vector<Msg>& ms, uint16_t type)
{
    for (const Msg& m : ms)
        if (m.type == type) return &m;
    return nullptr;
}

struct Link {
    std::string name;
    uint64_t addr;
};

void walk_group_btree(const Image& im, uint64_t node, uint64_t heap_data, uint64_t heap_size, std::vector<Link>& out, int depth, size_t& visited)
{
    if (depth > 32 || ++visited > (1u << 22)) throw NoVerdict{"group B-tree too deep (or cyclic)"};
    if (im.tag(node, "SNOD")) {
        const unsigned n = im.u16(node + 6);
        uint64_t p = node + 8;
        for (unsigned i = 0; i < n; i++, p += 40) {
            const uint64_t noff = im.u64(p);
            if (noff >= heap_size) throw NoVerdict{"link name outside the local heap"};
            const uint64_t s = heap_data + noff;
            const void* z = memchr(im.p + s, 0, (size_t)(heap_size - noff));
            if (!z) throw NoVerdict{"unterminated link name"};
            const uint64_t a = im.addr(p + 8);
            if (a == kUndef) throw NoVerdict{"undefined object address"};
            out.push_back({std::string((const char*)im.p + s, (const char*)z), a});
        }
        return;
    }
    if (!im.tag(node, "TREE") || im.u8(node + 4) != 0) throw NoVerdict{"bad group B-tree node"};
    const unsigned used = im.u16(node + 6);
    uint64_t p = node + 8 + 16;     // past the sibling addresses
    for (unsigned i = 0; i < used; i++) {
        p += 8;                     // key
        const uint64_t child = im.addr(p);
        if (child == kUndef) throw NoVerdict{"undefined B-tree child"};
        walk_group_btree(im, child, heap_data, heap_size, out, depth + 1, visited);
        p += 8;
    }
}

// the links of a group object: symbol-table storage, or compact link messages in its (version-1) header
void group_links(const Image& im, uint64_t at, std::vector<Link>& out)
{
    out.clear();
    std::vector<Msg> ms;
    header_messages(im, at, ms);
    if (const Msg* st = find_msg(ms, 0x0011)) {
        if (st->size < 16) throw NoVerdict{"short symbol-table message"};
        const uint64_t btree = im.addr(st->data), heap = im.addr(st->data + 8);
        if (btree == kUndef || heap == kUndef) throw NoVerdict{"undefined symbol table"};
        if (!im.tag(heap, "HEAP")) throw NoVerdict{"bad local heap"};
        const uint64_t heap_size = im.u64(heap + 8);
        const uint64_t heap_data = im.addr(heap + 24);
        if (heap_data == kUndef) throw NoVerdict{"undefined heap data"};
        im.need(heap_data, heap_size);
        size_t visited = 0;
        walk_group_btree(im, btree, heap_data, heap_size, out, 0, visited);
        return;
    }
    const Msg* info = find_msg(ms, 0x0002);
    bool any_link = false;
    for (const Msg& m : ms) any_link |= m.type == 0x0006;
    if (!info && !any_link) throw NoVerdict{"object is not a group"};
    if (info) {
        const uint8_t flags = im.u8(info->data + 1);
        const uint64_t q = info->data + 2 + ((flags & 1) ? 8 : 0);
        if (im.u64(q) != kUndef) throw NoVerdict{"dense (fractal-heap) group storage"};
    }
    for (const Msg& m : ms) {
        if (m.type != 0x0006) continue;
        const uint64_t end = m.data + m.size;
        if (im.u8(m.data) != 1) throw NoVerdict{"unknown link message version"};
        const uint8_t flags = im.u8(m.data + 1);
        uint64_t q = m.data + 2;
        unsigned ltype = 0;
        if (flags & 0x08) ltype = im.u8(q++);
        if (flags & 0x04) q += 8;
        if (flags & 0x10) q += 1;
        const unsigned lsz = 1u << (flags & 3);
        uint64_t nlen = 0;
        for (unsigned i = 0; i < lsz; i++) nlen |= (uint64_t)im.u8(q + i) << (8 * i);
        q += lsz;
        if (q > end || nlen > end - q) throw NoVerdict{"link name runs past its message"};
        im.need(q, nlen);
        std::string name((const char*)im.p + q, (size_t)nlen);
        q += nlen;
        if (ltype != 0) continue;   // soft / external links are not followed
        const uint64_t a = im.addr(q);
        if (a == kUndef) throw NoVerdict{"undefined object address"};
        out.push_back({std::move(name), a});
    }
}

bool find_link(const std::vector<Link>& ls, const char* name, uint64_t& addr)
{
    for (const Link& l : ls)
        if (l.name == name) {
            addr = l.addr;
            return true;
        }
    return false;
}

// a one-dimensional little-endian int16 dataset without filters: where its elements are
struct Signal {
    bool resolved = false;
    int64_t n = 0;
    int cls = 0;             // 0 compact, 1 contiguous, 2 chunked
    uint64_t where = kUndef; // compact / contiguous: offset of the data; chunked: the chunk B-tree's root
    uint64_t bytes = 0;      // compact: stored bytes
    uint32_t chunk_len = 0;
    int chunk_rank = 0;      // dimensionality in the layout message (= 2 for a 1-D dataset: the element size is the last dimension)
    struct Filter {
        uint16_t id;         // 1 deflate, 2 shuffle, 3 Fletcher-32
        uint32_t cd0;        // shuffle: bytes per element
    };
    std::vector<Filter> filters;     // in the order they were applied when the chunk was written
};

// HDF5's filter pipeline message (0x000B), versions 1 and 2 -> the filters this reader undoes
void parse_filters(const Image& im, const Msg& m, std::vector<Signal::Filter>& out)
{
    if (m.flags & 0x02) throw NoVerdict{"shared filter pipeline message"};
    const uint64_t end = m.data + m.size;
    const uint8_t ver = im.u8(m.data), nf = im.u8(m.data + 1);
    if (ver != 1 && ver != 2) throw NoVerdict{"unknown filter pipeline version"};
    if (nf > 32) throw NoVerdict{"more filters than a pipeline can hold"};
    uint64_t p = m.data + (ver == 1 ? 8 : 2);
    for (unsigned i = 0; i < nf; i++) {
        if (p + 8 > end) throw NoVerdict{"filter description runs past its message"};
        const uint16_t id = im.u16(p);
        p += 2;
        unsigned nlen = 0;
        if (ver == 1 || id >= 256) {
            nlen = im.u16(p);
            p += 2;
        }
        p += 2;                                  // flags (bit 0 "optional": how a failing filter is treated on WRITE)
        const unsigned ncv = im.u16(p);
        p += 2;
        p += ver == 1 ? ((nlen + 7u) & ~7u) : nlen;
        if (p > end || (uint64_t)ncv * 4 > end - p) throw NoVerdict{"filter description runs past its message"};
        const uint32_t cd0 = ncv ? im.u32(p) : 0;
        p += (uint64_t)ncv * 4 + ((ver == 1 && (ncv & 1)) ? 4 : 0);
        if (id == 1 || id == 3) out.push_back({id, cd0});
        else if (id == 2) {
            if (cd0 != 2) throw NoVerdict{"shuffle filter with an element size other than the signal's"};
            out.push_back({id, cd0});
        } else if (id == 32020) throw NoVerdict{"VBZ-compressed signal (needs ONT's HDF5 plugin)"};
        else throw NoVerdict{"signal stored with a filter other than deflate / shuffle / Fletcher-32"};
    }
}

// H5_checksum_fletcher32 (HDF5's variant: big-endian 16-bit words, one's-complement folding every 360 words)
uint32_t fletcher32(const uint8_t* d, size_t n)
{
    size_t len = n / 2;
    uint32_t s1 = 0, s2 = 0;
    while (len) {
        size_t t = len > 360 ? 360 : len;
        len -= t;
        do {
            s1 += ((uint32_t)d[0] << 8) | d[1];
            d += 2;
            s2 += s1;
        } while (--t);
        s1 = (s1 & 0xffff) + (s1 >> 16);
        s2 = (s2 & 0xffff) + (s2 >> 16);
    }
    if (n & 1) {
        s1 += (uint32_t)d[0] << 8;
        s2 += s1;
        s1 = (s1 & 0xffff) + (s1 >> 16);
        s2 = (s2 & 0xffff) + (s2 >> 16);
    }
    s1 = (s1 & 0xffff) + (s1 >> 16);
    s2 = (s2 & 0xffff) + (s2 >> 16);
    return (s2 << 16) | s1;
}

// One stored chunk back through the pipeline (last filter first; a set bit i of the chunk's mask = filter i was skipped when it was written)
// -> exactly chunk_len int16 in `out`.  a / b: scratch.
void unfilter_chunk(const Signal& s, const uint8_t* src, size_t n, uint32_t fmask, std::vector<uint8_t>& a, std::vector<uint8_t>& b, int16_t* out)
{
    const size_t raw = (size_t)s.chunk_len * 2;
    const uint8_t* cur = src;
    for (size_t i = s.filters.size(); i-- > 0;) {
        if ((fmask >> i) & 1u) continue;
        const Signal::Filter& f = s.filters[i];
        if (f.id == 3) {
            if (n < 4) throw NoVerdict{"checksummed chunk shorter than its checksum"};
            uint32_t stored;
            memcpy(&stored, cur + n - 4, 4);
            n -= 4;
            const uint32_t sum = fletcher32(cur, n);
            // (libhdf5 1.6.2 wrote the two 16-bit sums' bytes swapped; H5Z_filter_fletcher32 accepts either, so does this)
            const uint32_t swapped = ((sum & 0x00ff00ffu) << 8) | ((sum & 0xff00ff00u) >> 8);
            if (stored != sum && stored != swapped) throw NoVerdict{"chunk fails its Fletcher-32 checksum"};
        } else if (f.id == 1) {
            std::vector<uint8_t>& dst = (cur == a.data()) ? b : a;
            dst.resize(raw + 4 + 1);                       // (one chunk, perhaps its checksum; a byte more shows an over-long stream)
            uLongf got = (uLongf)dst.size();
            if (uncompress(dst.data(), &got, cur, (uLong)n) != Z_OK) throw NoVerdict{"chunk does not inflate"};
            cur = dst.data();
            n = (size_t)got;
        } else {                                           // shuffle: byte planes back into 2-byte elements (a trailing odd byte stays put)
            std::vector<uint8_t>& dst = (cur == a.data()) ? b : a;
            dst.resize(n);
            const size_t ne = n / 2;
            for (size_t k = 0; k < ne; k++) {
                dst[2 * k] = cur[k];
                dst[2 * k + 1] = cur[ne + k];
            }
            if (n & 1) dst[n - 1] = cur[n - 1];
            cur = dst.data();
        }
    }
    if (n != raw) throw NoVerdict{"a stored chunk does not decode to one chunk of samples"};
    memcpy(out, cur, raw);
}

// a fill value other than zero (fill-value message 0x0005, versions 1-3; the old message 0x0004): libhdf5 reads unallocated storage
// and missing chunks as that value, this reader as zero
bool nonzero_fill(const Image& im, const std::vector<Msg>& ms)
{
    auto any = [&](uint64_t p, uint64_t size, const Msg& m) {
        if (p > m.data + m.size || size > m.data + m.size - p) throw NoVerdict{"fill value runs past its message"};
        im.need(p, size);
        for (uint64_t i = 0; i < size; i++)
            if (im.p[p + i]) return true;
        return false;
    };
    if (const Msg* fv = find_msg(ms, 0x0005)) {
        if (fv->flags & 0x02) throw NoVerdict{"shared fill-value message"};
        const uint8_t ver = im.u8(fv->data);
        if (ver == 1 || ver == 2) {
            const bool defined = im.u8(fv->data + 3) != 0;
            if ((ver == 1 || defined) && fv->size >= 8) {
                const uint32_t size = im.u32(fv->data + 4);
                if (any(fv->data + 8, size, *fv)) return true;
            }
        } else if (ver == 3) {
            if (im.u8(fv->data + 1) & 0x20) {
                const uint32_t size = im.u32(fv->data + 2);
                if (any(fv->data + 6, size, *fv)) return true;
            }
        } else {
            throw NoVerdict{"unknown fill-value message version"};
        }
    }
    if (const Msg* old = find_msg(ms, 0x0004)) {
        if (old->flags & 0x02) throw NoVerdict{"shared fill-value message"};
        if (old->size >= 4 && any(old->data + 4, im.u32(old->data), *old)) return true;
    }
    return false;
}

void resolve_signal(const Image& im, uint64_t at, Signal& result)
{
    Signal s;                // (filled here, handed over only when everything about the dataset is understood: a failed attempt leaves `result` as it was)
    std::vector<Msg> ms;
    header_messages(im, at, ms);
    const Msg *dt = find_msg(ms, 0x0003), *sp = find_msg(ms, 0x0001), *lay = find_msg(ms, 0x0008);
    if (!dt || !sp || !lay) throw NoVerdict{"Signal is not a dataset"};
    const Msg* fl = find_msg(ms, 0x000B);
    if (fl) parse_filters(im, *fl, s.filters);
    if (find_msg(ms, 0x0007)) throw NoVerdict{"Signal lives in external files"};
    if (nonzero_fill(im, ms)) throw NoVerdict{"Signal has a fill value other than zero"};
    if ((dt->flags | sp->flags | lay->flags) & 0x02) throw NoVerdict{"shared (committed) datatype / dataspace message"};
    // datatype: class 0 (fixed point), little-endian, signed, 2 bytes, 16-bit precision at offset 0
    const uint8_t cv = im.u8(dt->data), bits0 = im.u8(dt->data + 1);
    if ((cv & 0x0f) != 0 || (bits0 & 1) || !(bits0 & 8) || im.u32(dt->data + 4) != 2) throw NoVerdict{"Signal is not little-endian int16"};
    if (dt->size >= 12 && (im.u16(dt->data + 8) != 0 || im.u16(dt->data + 10) != 16)) throw NoVerdict{"Signal is not plain int16"};
    // dataspace: rank 1
    const uint8_t sver = im.u8(sp->data), rank = im.u8(sp->data + 1);
    if ((sver != 1 && sver != 2) || rank != 1) throw NoVerdict{"Signal is not one-dimensional"};
    const uint64_t dim = im.u64(sp->data + (sver == 1 ? 8 : 4));
    if (dim > (uint64_t)1 << 40) throw NoVerdict{"implausible Signal length"};
    s.n = (int64_t)dim;
    const uint64_t p = lay->data;
    if (im.u8(p) != 3) throw NoVerdict{"data layout message is not version 3"};
    s.cls = im.u8(p + 1);
    if (s.cls == 1) {
        s.where = im.addr(p + 2);
        if (s.where != kUndef) im.need(s.where, dim * 2);
    } else if (s.cls == 2) {
        s.chunk_rank = im.u8(p + 2);
        if (s.chunk_rank != 2) throw NoVerdict{"chunked Signal is not one-dimensional"};
        s.where = im.addr(p + 3);
        s.chunk_len = im.u32(p + 11);
        if (im.u32(p + 15) != 2 || s.chunk_len == 0 || s.chunk_len > (1u << 28)) throw NoVerdict{"unexpected chunk geometry"};
    } else if (s.cls == 0) {
        s.bytes = im.u16(p + 2);
        s.where = p + 4;
        im.need(s.where, s.bytes);
    } else {
        throw NoVerdict{"unknown layout class"};
    }
    if (!s.filters.empty() && s.cls != 2) throw NoVerdict{"filters on a dataset that is not chunked"};
    if (!s.filters.empty() && s.chunk_len > (1u << 24)) throw NoVerdict{"implausibly large filtered chunk"};
    s.resolved = true;
    result = std::move(s);
}

struct Scratch {
    std::vector<uint8_t> a, b;
    std::vector<int16_t> chunk;
};

void copy_chunks(const Image& im, uint64_t node, const Signal& s, int16_t* out, int depth, size_t& visited, Scratch& sc)
{
    if (depth > 32 || ++visited > (1u << 22)) throw NoVerdict{"chunk B-tree too deep (or cyclic)"};
    if (!im.tag(node, "TREE") || im.u8(node + 4) != 1) throw NoVerdict{"bad chunk B-tree node"};
    const unsigned level = im.u8(node + 5), used = im.u16(node + 6);
    const uint64_t keysz = 8 + 8 * (uint64_t)s.chunk_rank;
    uint64_t p = node + 8 + 16;
    for (unsigned i = 0; i < used; i++, p += keysz + 8) {
        const uint32_t csize = im.u32(p), fmask = im.u32(p + 4);
        const uint64_t off0 = im.u64(p + 8);
        const uint64_t child = im.addr(p + keysz);
        if (child == kUndef) throw NoVerdict{"undefined chunk address"};
        if (level == 0) {
            if (off0 >= (uint64_t)s.n) continue;
            uint64_t cnt = std::min<uint64_t>(s.chunk_len, (uint64_t)s.n - off0);
            if (s.filters.empty()) {
                if (fmask) throw NoVerdict{"filtered chunk"};
                cnt = std::min<uint64_t>(cnt, csize / 2);
                im.need(child, cnt * 2);
                memcpy(out + off0, im.p + child, (size_t)cnt * 2);
            } else {
                im.need(child, csize);
                sc.chunk.resize(s.chunk_len);
                unfilter_chunk(s, im.p + child, csize, fmask, sc.a, sc.b, sc.chunk.data());
                memcpy(out + off0, sc.chunk.data(), (size_t)cnt * 2);
            }
        } else {
            copy_chunks(im, child, s, out, depth + 1, visited, sc);
        }
    }
}

void copy_signal(const Image& im, const Signal& s, int16_t* out, Scratch& sc)
{
    if (s.n == 0) return;
    if (s.cls == 1) {
        if (s.where == kUndef) memset(out, 0, (size_t)s.n * 2);     // never written: the fill value (zero)
        else memcpy(out, im.p + s.where, (size_t)s.n * 2);
    } else if (s.cls == 0) {
        memset(out, 0, (size_t)s.n * 2);
        memcpy(out, im.p + s.where, (size_t)std::min<uint64_t>((uint64_t)s.n * 2, s.bytes));
    } else {
        memset(out, 0, (size_t)s.n * 2);                            // chunks that were never written read as the fill value
        if (s.where != kUndef) {
            size_t visited = 0;
            copy_chunks(im, s.where, s, out, 0, visited, sc);
        }
    }
}

// a string attribute of an object -- fixed-length, or variable-length out of the global heap (version 1-3 attribute messages); false: the object has no such attribute
bool string_attr(const Image& im, uint64_t at, const char* name, std::string& out)
{
    std::vector<Msg> ms;
    header_messages(im, at, ms);
    if (const Msg* ai = find_msg(ms, 0x0015)) {      // attribute info: attributes may live in a fractal heap ("dense" storage) instead of the header
        const uint8_t fl = im.u8(ai->data + 1);
        if (im.u64(ai->data + 2 + ((fl & 1) ? 2 : 0)) != kUndef) throw NoVerdict{"dense attribute storage"};
    }
    for (const Msg& m : ms) {
        if (m.type != 0x000C) continue;
        if (m.flags & 0x02) throw NoVerdict{"shared attribute message"};
        const uint8_t ver = im.u8(m.data);
        if (ver < 1 || ver > 3) continue;
        const unsigned nsz = im.u16(m.data + 2), tsz = im.u16(m.data + 4), ssz = im.u16(m.data + 6);
        uint64_t p = m.data + 8 + (ver == 3 ? 1 : 0);
        auto pad = [&](unsigned x) { return ver == 1 ? (uint64_t)((x + 7) & ~7u) : (uint64_t)x; };
        im.need(p, nsz);
        const size_t nl = strnlen((const char*)im.p + p, nsz);
        const bool mine = nl == strlen(name) && memcmp(im.p + p, name, nl) == 0;
        p += pad(nsz);
        const uint64_t tp = p;
        p += pad(tsz);
        p += pad(ssz);
        if (!mine) continue;
        if (ver >= 2 && (im.u8(m.data + 1) & 0x03)) throw NoVerdict{"read_id attribute with a shared (committed) datatype / dataspace"};
        const uint8_t cls = im.u8(tp) & 0x0f;
        if (cls == 9) {      // variable-length string (what h5py writes for a Python str: ont_fast5_api's multi_to_single output): the value is in the global heap
            if ((im.u8(tp + 1) & 0x0f) != 1) throw NoVerdict{"read_id attribute is a variable-length sequence, not a string"};
            if (p > m.data + m.size || 16 > m.data + m.size - p) throw NoVerdict{"attribute value runs past its message"};
            const uint32_t len = im.u32(p), idx = im.u32(p + 12);
            const uint64_t col = im.addr(p + 4);
            if (col == kUndef || len == 0) {      // (a null / empty string)
                out.clear();
                return true;
            }
            if (!im.tag(col, "GCOL") || im.u8(col + 4) != 1) throw NoVerdict{"bad global heap collection"};
            const uint64_t csize = im.u64(col + 8);
            im.need(col, csize);
            for (uint64_t q = col + 16; q + 16 <= col + csize;) {
                const unsigned oi = im.u16(q);
                const uint64_t osz = im.u64(q + 8);
                if (oi == 0) break;               // (the collection's free space)
                if (osz > col + csize - (q + 16)) throw NoVerdict{"global heap object runs past its collection"};
                if (oi == idx) {
                    if (len > osz) throw NoVerdict{"variable-length string longer than its heap object"};
                    out.assign((const char*)im.p + q + 16, strnlen((const char*)im.p + q + 16, len));
                    return true;
                }
                q += 16 + ((osz + 7) & ~(uint64_t)7);
            }
            throw NoVerdict{"variable-length string not found in its heap collection"};
        }
        if (cls != 3) throw NoVerdict{"read_id attribute is not a string"};
        if ((im.u8(tp + 1) & 0x0f) > 1) throw NoVerdict{"read_id attribute is a space-padded string"};   // (0 NUL-terminated, 1 NUL-padded: what strnlen below undoes)
        const uint32_t size = im.u32(tp + 4);
        if (p > m.data + m.size || size > m.data + m.size - p) throw NoVerdict{"attribute value runs past its message"};
        im.need(p, size);
        out.assign((const char*)im.p + p, strnlen((const char*)im.p + p, size));
        return true;
    }
    return false;
}

}  // namespace

struct rd_fast5 {
    Image im;
    void* map = nullptr;     // the mapping this handle owns (rd_fast5_open), or null (rd_fast5_open_mem)
    size_t map_len = 0;
    struct Entry {
        std::string id;
        uint64_t group;      // object header of the read's group
        bool multi;
        bool id_known;
        Signal sig;
    };
    std::vector<Entry> reads;
};

namespace {

void index_file(rd_fast5* f)
{
    Image& im = f->im;
    // superblock: the signature at 0, 512, 1024, ...
    uint64_t sb = kUndef;
    for (uint64_t off = 0; off + 8 <= im.n; off = off == 0 ? 512 : off * 2)
        if (memcmp(im.p + off, kSig, 8) == 0) {
            sb = off;
            break;
        }
    if (sb == kUndef) throw NoVerdict{"not an HDF5 file"};
    const uint8_t ver = im.u8(sb + 8);
    if (ver > 1) throw NoVerdict{"superblock version 2 or later"};
    if (im.u8(sb + 13) != 8 || im.u8(sb + 14) != 8) throw NoVerdict{"offsets / lengths are not 8 bytes"};
    uint64_t p = sb + 24 + (ver == 1 ? 4 : 0);
    im.base = 0;
    const uint64_t base = im.u64(p);
    if (base > im.n) throw NoVerdict{"base address outside the file"};
    im.base = base;
    p += 8 * 4;              // base, free-space, end-of-file, driver-information addresses
    const uint64_t root = im.addr(p + 8);
    if (root == kUndef) throw NoVerdict{"undefined root group"};
    std::vector<Link> links, sub;
    group_links(im, root, links);
    std::sort(links.begin(), links.end(), [](const Link& a, const Link& b) { return a.name < b.name; });
    bool multi = false;
    for (const Link& l : links) multi |= l.name.compare(0, 5, "read_") == 0;
    if (multi) {
        for (const Link& l : links)
            if (l.name.compare(0, 5, "read_") == 0) f->reads.push_back({l.name.substr(5), l.addr, true, true, Signal{}});
        return;
    }
    uint64_t raw, rr;
    if (!find_link(links, "Raw", raw)) return;        // neither layout: no reads (fast5.iter_reads yields nothing either)
    group_links(im, raw, sub);
    if (!find_link(sub, "Reads", rr)) return;
    group_links(im, rr, links);
    std::sort(links.begin(), links.end(), [](const Link& a, const Link& b) { return a.name < b.name; });
    for (const Link& l : links) f->reads.push_back({l.name, l.addr, false, false, Signal{}});
}

void resolve_entry(rd_fast5* f, rd_fast5::Entry& e)
{
    if (e.sig.resolved) return;
    std::vector<Link> ls;
    uint64_t sig;
    if (e.multi) {
        uint64_t raw;
        group_links(f->im, e.group, ls);
        if (!find_link(ls, "Raw", raw)) throw NoVerdict{"read group without Raw"};
        group_links(f->im, raw, ls);
        if (!find_link(ls, "Signal", sig)) throw NoVerdict{"read group without Raw/Signal"};
    } else {
        group_links(f->im, e.group, ls);
        if (!find_link(ls, "Signal", sig)) throw NoVerdict{"read group without Signal"};
        if (!e.id_known) {
            std::string rid;
            if (string_attr(f->im, e.group, "read_id", rid)) e.id = rid;
            e.id_known = true;
        }
    }
    resolve_signal(f->im, sig, e.sig);
}

template <typename F>
int guarded(const char* what, F&& fn)
{
    try {
        return fn();
    } catch (const NoVerdict& e) {
        rd_set_error("%s: %s (not the classic layout this reader walks: use libhdf5)", what, e.why);
        return RD_ERR_FORMAT;
    } catch (const std::bad_alloc&) {
        rd_set_error("%s: out of host memory", what);
        return RD_ERR_NOMEM;
    }
}

}  // namespace

extern "C" int rd_fast5_open_mem(const void* buf, size_t n, rd_fast5** out)
{
    RD_REQUIRE(out && (buf || n == 0), "rd_fast5_open_mem: null argument");
    *out = nullptr;
    return guarded("rd_fast5_open_mem", [&]() {
        rd_fast5* f = new rd_fast5();
        f->im.p = (const uint8_t*)buf;
        f->im.n = n;
        try {
            index_file(f);
        } catch (...) {
            delete f;
            throw;
        }
        *out = f;
        return RD_OK;
    });
}

extern "C" int rd_fast5_open(const char* path, rd_fast5** out)
{
    RD_REQUIRE(path && out, "rd_fast5_open: null argument");
    *out = nullptr;
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) {
        rd_set_error("rd_fast5_open: cannot open %s: %s", path, strerror(errno));
        return RD_ERR_IO;
    }
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size <= 0) {
        close(fd);
        rd_set_error("rd_fast5_open: %s is empty or cannot be examined", path);
        return RD_ERR_FORMAT;
    }
    void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) {
        rd_set_error("rd_fast5_open: cannot map %s: %s", path, strerror(errno));
        return RD_ERR_IO;
    }
    (void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
    int rc = rd_fast5_open_mem(m, (size_t)st.st_size, out);
    if (rc != RD_OK) {
        munmap(m, (size_t)st.st_size);
        return rc;
    }
    (*out)->map = m;
    (*out)->map_len = (size_t)st.st_size;
    return RD_OK;
}

extern "C" void rd_fast5_close(rd_fast5* f)
{
    if (!f) return;
    if (f->map) munmap(f->map, f->map_len);
    delete f;
}

extern "C" int rd_fast5_count(const rd_fast5* f, int64_t* n_reads)
{
    RD_REQUIRE(f && n_reads, "rd_fast5_count: null argument");
    *n_reads = (int64_t)f->reads.size();
    return RD_OK;
}

extern "C" int rd_fast5_lengths(rd_fast5* f, int64_t lo, int64_t hi, int64_t* n_samples)
{
    RD_REQUIRE(f && n_samples, "rd_fast5_lengths: null argument");
    RD_REQUIRE(lo >= 0 && lo <= hi && hi <= (int64_t)f->reads.size(), "rd_fast5_lengths: reads [%lld, %lld) of %zu", (long long)lo, (long long)hi,
               f->reads.size());
    return guarded("rd_fast5_lengths", [&]() {
        for (int64_t i = lo; i < hi; i++) {
            resolve_entry(f, f->reads[(size_t)i]);
            n_samples[i - lo] = f->reads[(size_t)i].sig.n;
        }
        return RD_OK;
    });
}

extern "C" int rd_fast5_read_batch(rd_fast5* f, int64_t lo, int64_t hi, int16_t* samples, int64_t cap, int64_t* offsets, char* ids, int id_stride)
{
    RD_REQUIRE(f && offsets && (samples || cap == 0), "rd_fast5_read_batch: null argument");
    RD_REQUIRE(lo >= 0 && lo <= hi && hi <= (int64_t)f->reads.size(), "rd_fast5_read_batch: reads [%lld, %lld) of %zu", (long long)lo, (long long)hi,
               f->reads.size());
    RD_REQUIRE(!ids || id_stride >= 2, "rd_fast5_read_batch: id_stride %d", id_stride);
    return guarded("rd_fast5_read_batch", [&]() {
        int64_t at = 0;
        Scratch sc;
        for (int64_t i = lo; i < hi; i++) {
            rd_fast5::Entry& e = f->reads[(size_t)i];
            resolve_entry(f, e);
            offsets[i - lo] = at;
            if (e.sig.n > cap - at) {
                rd_set_error("rd_fast5_read_batch: the block needs more than %lld samples (size it with rd_fast5_lengths)", (long long)cap);
                return RD_ERR_ARG;
            }
            copy_signal(f->im, e.sig, samples + at, sc);
            at += e.sig.n;
            if (ids) {
                if (e.id.size() + 1 > (size_t)id_stride || memchr(e.id.data(), 0, e.id.size())) throw NoVerdict{"read id does not fit the id block"};
                memcpy(ids + (size_t)(i - lo) * id_stride, e.id.c_str(), e.id.size() + 1);
            }
        }
        offsets[hi - lo] = at;
        return RD_OK;
    });
}
