// gfx950 kernels of the batched zstd decoder.
//
// Stage map (reference routine each stage must reproduce, file:line in the reference tree):
//   k_literals   HUF_readStats_body (common/entropy_common.pas:311-392), FSE_decompress_wksp_body
//                (common/fse_decompress.pas:341-383), HUF_readDTableX1_wksp (decompress/huf_decompress.pas:133-285),
//                HUF_decompress1X1/4X1_usingDTable_internal_body (:350-559)
//   k_sequences  FSE_readNCount_body (common/entropy_common.pas:83-256), ZSTD_buildFSETable_body /
//                ZSTD_buildSeqTable (decompress/zstd_decompress_block.pas:436-688), ZSTD_decodeSequence (:1083-1230)
//   k_layout     the op/rep hand-over between blocks of ZSTD_decompressFrame (decompress/zstd_decompressf.pas:772-808)
//   k_execute    ZSTD_execSequence / ZSTD_execSequenceEnd (decompress/zstd_decompress_block.pas:896-1039), last
//                literals (:1308-1317), ZSTD_copyRawBlock / ZSTD_setRleBlock (decompress/zstd_decompressf.pas:699-731)
//   k_checksum   XXH64 (common/xxhash.pas:393-488) + the frame verdict of ZSTD_decompressFrame (:810-828)
//   k_digest_dict ZSTD_loadDEntropy (decompress/zstd_decompressf.pas:1261-1356)
//
// All arithmetic is integer; wave = 64 lanes; every workgroup is a single wave, so lanes of a
// workgroup run in lockstep and exchange data through LDS with a fence only.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include "zstdgpu_internal.h"

namespace zg {
namespace {

#define DEVI __device__ __forceinline__

// ---------------------------------------------------------------- small helpers
DEVI uint32_t hb32(uint32_t v) { return 31u - (uint32_t)__builtin_clz(v); }

// Job counts of a batch: launch parameters, or -- for a batch whose descriptors never left the device -- the device
// totals (none when the batch does not fit its scratch).
DEVI bool batch_overflow(const BatchParams& p) { return p.devTotals && p.devTotals[7] != 0; }
DEVI uint32_t n_lit_jobs(const BatchParams& p) { return p.devTotals ? (p.devTotals[7] ? 0u : (uint32_t)p.devTotals[4]) : p.nLitJobs; }
DEVI uint32_t n_seq_jobs(const BatchParams& p) { return p.devTotals ? (p.devTotals[7] ? 0u : (uint32_t)p.devTotals[5]) : p.nSeqJobs; }

// orders LDS and global accesses of the lanes of this (single-wave) workgroup
DEVI void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// LDS-only ordering between the lanes of this wave: LDS operations of a wave complete in issue order,
// so draining lgkmcnt is enough; outstanding global loads and stores are deliberately not waited for.
DEVI void lds_sync()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// Cross-lane moves without LDS: DPP row shifts inside rows of 16 lanes and row broadcasts between them.
template <int CTRL, int ROWMASK>
DEVI uint32_t dpp_or0(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROWMASK, 0xF, false); }
// inclusive prefix sum over the 64 lanes of a wave (all lanes active)
DEVI uint32_t wave_incl_scan(uint32_t v)
{
    v += dpp_or0<0x111, 0xF>(v); v += dpp_or0<0x112, 0xF>(v); v += dpp_or0<0x114, 0xF>(v); v += dpp_or0<0x118, 0xF>(v);   // row_shr 1, 2, 4, 8
    v += dpp_or0<0x142, 0xA>(v);      // lane 15 of rows 0 and 2 onto rows 1 and 3
    v += dpp_or0<0x143, 0xC>(v);      // lane 31 onto rows 2 and 3
    return v;
}
// value of the next higher lane (own value in lane 63)
DEVI uint32_t wave_next(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x130, 0xF, 0xF, false); }

DEVI uint64_t ld_u64(const uint8_t* p)
{
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}
// little-endian 32-bit read at byte index i of a buffer of `size` bytes, zero beyond the end
template <typename SrcT>
DEVI uint32_t rd32z(SrcT p, uint32_t size, uint32_t i)
{
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (i + k < size) v |= (uint32_t)p[i + k] << (8 * k);
    return v;
}
// (one unaligned access when all four bytes exist)
DEVI uint32_t rd32z(const uint8_t* p, uint32_t size, uint32_t i)
{
    if (i + 4 <= size) { uint32_t v; __builtin_memcpy(&v, p + i, 4); return v; }
    return rd32z<const uint8_t*>(p, size, i);
}

// Bytes of a table description: the first nStaged of them were copied to LDS, the rest (rare) are read from memory.
struct HdrBytes {
    const uint8_t* lds;
    const uint8_t* mem;
    uint32_t nStaged;
    DEVI uint8_t operator[](uint32_t i) const { return i < nStaged ? lds[i] : mem[i]; }
};
DEVI uint32_t rd32z(HdrBytes const& p, uint32_t size, uint32_t i)
{
    if (i + 4 <= size && i + 4 <= p.nStaged) { uint32_t v; __builtin_memcpy(&v, p.lds + i, 4); return v; }
    return rd32z<HdrBytes const&>(p, size, i);
}

// Backward bit reader (BIT_DStream_t, common/bitstream.pas:40-51, :206-398) kept as a logical bit
// position: `left()` bits of the stream remain unread.  The 64-bit window is re-read from global
// memory at byte granularity so that at most 7 bits of it are already consumed.
struct BitR {
    const uint8_t* start;
    uint64_t w;
    int c;      // bits of w consumed
    int base;   // left() + c, constant between refills

    DEVI void refill_at(int bitsLeft)
    {
        int const byteIdx = (bitsLeft - 57) >> 3;   // may be -8..-1 near the stream start: frame/block headers precede every stream
        w = ld_u64(start + byteIdx);
        c = 8 * byteIdx + 64 - bitsLeft;
        base = bitsLeft + c;
    }
    // BIT_initDStream: false when the stream is empty or its last byte carries no end mark
    DEVI bool init(const uint8_t* s, uint32_t size)
    {
        start = s; w = 0; c = 0; base = -1;
        if (size == 0) return false;
        uint32_t const last = s[size - 1];
        if (last == 0) return false;
        refill_at((int)(size * 8) - (int)(8 - hb32(last)));
        return true;
    }
    DEVI int left() const { return base - c; }
    DEVI void refill()
    {
        int const l = left();
        if (l >= 0) refill_at(l);
    }
    // read n <= 32 bits; caller guarantees c + n <= 64 on valid streams (garbage, but in range, otherwise)
    DEVI uint32_t read(uint32_t n)
    {
        uint32_t const v = (uint32_t)(((w << (c & 63)) >> 1) >> (63 - n));
        c += (int)n;
        return v;
    }
};

// Bit cursor of the two hot loops.  The loops never touch global memory: a lane group first stages the
// top of the unread stream into an LDS window (aligned 16-byte units, coalesced), then one lane
// decodes a bounded number of symbols from that window.  (Vector-memory loads and stores inside the
// loop would serialise it, because a wave's loads and stores retire in issue order.)
// The cursor Q counts the window bits below the next unread bit; a 64-bit peek at Q takes three
// window dwords and two v_alignbit, fields come out with v_bfe, and nothing is renormalised.
DEVI void peek64(const uint32_t* win, uint32_t Q, uint32_t& tHi, uint32_t& tLo)
{
    uint32_t const d = Q >> 5, sft = Q & 31;
    uint32_t const a0 = win[d - 2], a1 = win[d - 1], a2 = win[d];
    tHi = __builtin_amdgcn_alignbit(a2, a1, sft);
    tLo = __builtin_amdgcn_alignbit(a1, a0, sft);
}
// top 32 bits of ({hi,lo} << n), n in [0,63]
DEVI uint32_t shl_hi(uint32_t hi, uint32_t lo, uint32_t n) { return (uint32_t)(((((uint64_t)hi << 32) | lo) << n) >> 32); }
// `n` bits of t that start `skip` bits below its top: skip + n <= 32, n == 0 gives 0
DEVI uint32_t take(uint32_t t, uint32_t skip, uint32_t n) { return __builtin_amdgcn_ubfe(t, 32 - skip - n, n); }

// 16 bytes at `addr` (16-byte aligned), clamped to `floor` (the source buffer's first byte) so a window
// that reaches below the buffer reads the buffer's first bytes instead; such bytes lie below the
// stream start and are never legitimately consumed.  Upward a unit ends at most 15 bytes past the last
// byte of a stream, inside the same 16-byte line of the source buffer's last page.
DEVI uint4 ld_unit(uintptr_t addr, uintptr_t floor)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) u32x4* gptr;     // global memory: keeps the load off the flat path
    u32x4 const x = *(gptr)(addr >= floor ? addr : floor);
    return make_uint4(x.x, x.y, x.z, x.w);
}

// the same at any byte address
DEVI uint4 ld_unit_any(uintptr_t addr, uintptr_t floor)
{
    typedef uint32_t u32x4a1 __attribute__((ext_vector_type(4), aligned(1)));
    typedef const __attribute__((address_space(1))) u32x4a1* gptr;
    u32x4a1 const x = *(gptr)(addr >= floor ? addr : floor);
    return make_uint4(x.x, x.y, x.z, x.w);
}

// n <= 256 bytes from global memory into LDS (16-byte aligned there) by a group of 16 lanes: one 16-byte load per
// lane, all in flight together; the last, partial chunk is loaded as the 16 bytes that end with byte n - 1.
DEVI void grp_stage_bytes(uint8_t* dst, const uint8_t* src, uint32_t n, uint32_t gl)
{
    typedef uint32_t u32x4a1 __attribute__((ext_vector_type(4), aligned(1)));
    uint32_t const full = n >> 4;
    if (gl < full) {
        u32x4a1 const x = *(const u32x4a1*)(src + 16 * gl);
        *(uint4*)(dst + 16 * gl) = make_uint4(x.x, x.y, x.z, x.w);
    } else if (gl == full && (n & 15)) {
        if (n >= 16) {
            u32x4a1 const x = *(const u32x4a1*)(src + n - 16);
            __builtin_memcpy(dst + n - 16, &x, 16);
        } else {
            for (uint32_t i = 0; i < n; i++) dst[i] = src[i];
        }
    }
}

// n copies of a 1-byte or 2-byte value at dst (LDS, any alignment) by one lane: 16 bytes per store, the tail as a store
// that ends with the last byte (a dominant symbol owns hundreds of cells: byte by byte, its lane would keep the group waiting)
DEVI void fill_bytes(uint8_t* dst, uint32_t v32, uint32_t nBytes)
{
    uint4 const x = make_uint4(v32, v32, v32, v32);
    if (nBytes >= 16) {
        for (uint32_t i = 0; i + 16 <= nBytes; i += 16) __builtin_memcpy(dst + i, &x, 16);
        // (for 2-byte values nBytes and every offset used are even, so the pattern stays in phase)
        if (nBytes & 15) __builtin_memcpy(dst + nBytes - 16, &x, 16);
    } else {
        if (nBytes & 8) { __builtin_memcpy(dst, &x, 8); dst += 8; }
        if (nBytes & 4) { __builtin_memcpy(dst, &x, 4); dst += 4; }
        if (nBytes & 2) { __builtin_memcpy(dst, &x, 2); dst += 2; }
        if (nBytes & 1) dst[0] = (uint8_t)v32;
    }
}

// ---------------------------------------------------------------- FSE_readNCount (entropy_common.pas:83-256)
constexpr uint32_t kFail = 0xFFFFFFFFu;

// norm[] receives maxSV+1 entries.  Returns the header size in bytes or kFail.
template <typename NormPtr, typename SrcT>
DEVI uint32_t dev_readNCount(NormPtr norm, uint32_t& maxSV, uint32_t& tableLog, SrcT src, uint32_t hbSize)
{
    int const iend = hbSize < 8 ? 8 : (int)hbSize;    // short headers are decoded from a zero-padded copy (:100-113)
    int ip = 0;
    uint32_t const maxSV1 = maxSV + 1;
    for (uint32_t s = 0; s < maxSV1; s++) norm[s] = 0;
    uint32_t bitStream = rd32z(src, hbSize, 0);
    int nbBits = (int)(bitStream & 0xF) + 5;
    if (nbBits > 15) return kFail;
    bitStream >>= 4;
    int bitCount = 4;
    tableLog = (uint32_t)nbBits;
    int remaining = (1 << nbBits) + 1;
    int threshold = 1 << nbBits;
    nbBits++;
    uint32_t charnum = 0;
    bool previous0 = false;
    for (;;) {
        if (previous0) {
            int repeats = (int)(__builtin_ctz(~bitStream | 0x80000000u) >> 1);
            while (repeats >= 12) {
                charnum += 3 * 12;
                if (ip <= iend - 7) ip += 3;
                else { bitCount -= 8 * (iend - 7 - ip); bitCount &= 31; ip = iend - 4; }
                bitStream = rd32z(src, hbSize, (uint32_t)ip) >> bitCount;
                repeats = (int)(__builtin_ctz(~bitStream | 0x80000000u) >> 1);
                if (charnum >= maxSV1) break;   // already an error; leave the inner loop early
            }
            charnum += 3 * (uint32_t)repeats;
            bitStream >>= 2 * repeats;
            bitCount += 2 * repeats;
            charnum += bitStream & 3;
            bitCount += 2;
            if (charnum >= maxSV1) break;
            if ((ip <= iend - 7) || (ip + (bitCount >> 3) <= iend - 4)) { ip += bitCount >> 3; bitCount &= 7; }
            else { bitCount -= 8 * (iend - 4 - ip); bitCount &= 31; ip = iend - 4; }
            bitStream = rd32z(src, hbSize, (uint32_t)ip) >> bitCount;
        }
        {
            int const max = (2 * threshold - 1) - remaining;
            int count;
            if ((bitStream & (uint32_t)(threshold - 1)) < (uint32_t)max) { count = (int)(bitStream & (uint32_t)(threshold - 1)); bitCount += nbBits - 1; }
            else { count = (int)(bitStream & (uint32_t)(2 * threshold - 1)); if (count >= threshold) count -= max; bitCount += nbBits; }
            count--;
            if (count >= 0) remaining -= count; else remaining += count;
            norm[charnum++] = (int16_t)count;
            previous0 = (count == 0);
            if (remaining < threshold) {
                if (remaining <= 1) break;
                nbBits = (int)hb32((uint32_t)remaining) + 1;
                threshold = 1 << (nbBits - 1);
            }
            if (charnum >= maxSV1) break;
            if ((ip <= iend - 7) || (ip + (bitCount >> 3) <= iend - 4)) { ip += bitCount >> 3; bitCount &= 7; }
            else { bitCount -= 8 * (iend - 4 - ip); bitCount &= 31; ip = iend - 4; }
            bitStream = rd32z(src, hbSize, (uint32_t)ip) >> bitCount;
        }
    }
    if (remaining != 1) return kFail;
    if (charnum > maxSV1) return kFail;
    if (bitCount > 32) return kFail;
    maxSV = charnum - 1;
    ip += (bitCount + 7) >> 3;
    if ((uint32_t)ip > hbSize) return kFail;
    return (uint32_t)ip;
}

// ---------------------------------------------------------------- Huffman weights
// Scratch layout inside a job's (still unused) table area while the weights are decoded.
constexpr uint32_t kWNormOff = 0;       // int16 norm[256]
constexpr uint32_t kWNextOff = 512;     // uint16 symbolNext[256]
constexpr uint32_t kWTabOff = 1024;     // uint32 dtable[64]  {newState<<16 | nbBits<<8 | symbol}

// FSE_decompress_wksp_body (fse_decompress.pas:341-383) for the <=255 Huffman weights: header, table
// (FSE_buildDTable_internal, :40-176), two-state decode (FSE_decompress_usingDTable_generic, :242-316).
// Returns the number of weights or kFail.
template <typename BytePtr>
DEVI uint32_t dev_fse_weights(BytePtr weights, uint32_t maxOut, const uint8_t* src, uint32_t srcSize, uint8_t* scratch)
{
    int16_t* const norm = (int16_t*)(scratch + kWNormOff);
    uint16_t* const symNext = (uint16_t*)(scratch + kWNextOff);
    uint32_t* const dt = (uint32_t*)(scratch + kWTabOff);
    uint32_t maxSV = 255, tableLog = 0;
    uint32_t const hSize = dev_readNCount(norm, maxSV, tableLog, src, srcSize);
    if (hSize == kFail) return kFail;
    if (tableLog > 6) return kFail;
    uint32_t const tableSize = 1u << tableLog;
    uint32_t highThreshold = tableSize - 1;
    for (uint32_t s = 0; s <= maxSV; s++) {
        if (norm[s] == -1) { dt[highThreshold--] = s; symNext[s] = 1; }
        else symNext[s] = (uint16_t)norm[s];
    }
    {
        uint32_t const mask = tableSize - 1, step = (tableSize >> 1) + (tableSize >> 3) + 3;
        uint32_t pos = 0;
        for (uint32_t s = 0; s <= maxSV; s++) {
            int const n = norm[s];
            for (int i = 0; i < n; i++) {
                dt[pos] = s;
                pos = (pos + step) & mask;
                while (pos > highThreshold) pos = (pos + step) & mask;
            }
        }
        if (pos != 0) return kFail;
    }
    for (uint32_t u = 0; u < tableSize; u++) {
        uint32_t const sym = dt[u];
        uint32_t const ns = symNext[sym]++;
        uint32_t const nb = tableLog - hb32(ns);
        dt[u] = (((ns << nb) - tableSize) << 16) | (nb << 8) | sym;
    }
    // two-state decode.  `cur` = bits of the stream not yet read; every read looks at the 32 bits below the cursor
    // (an 8-byte unaligned load that may start up to 4 bytes before the stream: other header bytes always precede it).
    const uint8_t* const st = src + hSize;
    uint32_t const stSize = srcSize - hSize;
    if (stSize == 0) return kFail;                      // BIT_initDStream: empty stream
    uint32_t const lastByte = st[stSize - 1];
    if (lastByte == 0) return kFail;                    // no end mark
    int cur = (int)(stSize * 8) - (int)(8 - hb32(lastByte));
    auto read = [&](uint32_t nb) -> uint32_t {
        int const bo = cur - 32;
        uint32_t const t = (uint32_t)(ld_u64(st + (bo >> 3)) >> (bo & 7));
        cur -= (int)nb;
        return __builtin_amdgcn_ubfe(t, 32 - nb, nb);   // garbage once the cursor is below zero; callers stop there
    };
    uint32_t s1 = read(tableLog);
    uint32_t s2 = read(tableLog);
    if (cur < 0) return kFail;   // states alone overran the stream: upstream decodes garbage here, never a valid header
    uint32_t n = 0;
    for (;;) {
        if (n + 2 > maxOut) return kFail;
        { uint32_t const d = dt[s1]; weights[n++] = (uint8_t)d; s1 = (d >> 16) + read((d >> 8) & 0xFF); }
        if (cur < 0) { weights[n++] = (uint8_t)dt[s2]; break; }
        if (n + 2 > maxOut) return kFail;
        { uint32_t const d = dt[s2]; weights[n++] = (uint8_t)d; s2 = (d >> 16) + read((d >> 8) & 0xFF); }
        if (cur < 0) { weights[n++] = (uint8_t)dt[s1]; break; }
    }
    return n;
}

// HUF_readStats_body (entropy_common.pas:311-392) followed by the rank scan of
// HUF_readDTableX1_wksp (huf_decompress.pas:133-285): fills weights[0..nbSymbols) and pos[n] = first
// table cell of symbol n.  Returns header bytes consumed or kFail.  Runs on one lane.
template <typename BytePtr, typename PosPtr, typename RankPtr>
DEVI uint32_t dev_huf_read_table_desc(const uint8_t* src, uint32_t srcSize, BytePtr weights, PosPtr pos, RankPtr rank,
                                      uint8_t* scratch, uint32_t& nbSymbols, uint32_t& tableLog)
{
    if (srcSize == 0) return kFail;
    uint32_t iSize = src[0];
    uint32_t oSize;
    if (iSize >= 128) {
        oSize = iSize - 127;
        iSize = (oSize + 1) / 2;
        if (iSize + 1 > srcSize) return kFail;
        for (uint32_t n = 0; n < oSize; n += 2) {
            uint32_t const b = src[1 + n / 2];
            weights[n] = (uint8_t)(b >> 4);
            weights[n + 1] = (uint8_t)(b & 15);
        }
    } else {
        if (iSize + 1 > srcSize) return kFail;
        oSize = dev_fse_weights(weights, 255, src + 1, iSize, scratch);
        if (oSize == kFail) return kFail;
    }
    for (int r = 0; r <= 12; r++) rank[r] = 0;
    uint32_t weightTotal = 0;
    for (uint32_t n = 0; n < oSize; n++) {
        uint32_t const w = weights[n];
        if (w >= 12) return kFail;
        rank[w] = rank[w] + 1;
        weightTotal += (1u << w) >> 1;
    }
    if (weightTotal == 0) return kFail;
    tableLog = hb32(weightTotal) + 1;
    if (tableLog > 12) return kFail;
    {
        uint32_t const rest = (1u << tableLog) - weightTotal;
        uint32_t const lastWeight = hb32(rest) + 1;
        if ((1u << hb32(rest)) != rest) return kFail;
        weights[oSize] = (uint8_t)lastWeight;
        rank[lastWeight] = rank[lastWeight] + 1;
    }
    if (rank[1] < 2 || (rank[1] & 1)) return kFail;
    nbSymbols = oSize + 1;
    {   // rank[w] := first cell of weight class w
        uint32_t next = 0;
        for (uint32_t n = 1; n < tableLog + 1; n++) { uint32_t const cur = next; next += rank[n] << (n - 1); rank[n] = cur; }
    }
    for (uint32_t n = 0; n < nbSymbols; n++) {
        uint32_t const w = weights[n];
        uint32_t const len = (1u << w) >> 1;
        pos[n] = (uint16_t)rank[w];
        rank[w] = rank[w] + len;
    }
    return iSize + 1;
}

// First half of HUF_readStats_body (entropy_common.pas:311-349): the weights as listed in the description, direct
// or FSE-compressed.  One lane.  Returns the description's size in bytes or kFail; nListed = weights listed.
template <typename BytePtr>
DEVI uint32_t dev_huf_listed_weights(const uint8_t* src, uint32_t srcSize, BytePtr weights, uint8_t* scratch, uint32_t& nListed)
{
    if (srcSize == 0) return kFail;
    uint32_t iSize = src[0];
    if (iSize >= 128) {
        nListed = iSize - 127;
        iSize = (nListed + 1) / 2;
        if (iSize + 1 > srcSize) return kFail;
        for (uint32_t n = 0; n < nListed; n += 2) {
            uint32_t const b = src[1 + n / 2];
            weights[n] = (uint8_t)(b >> 4);
            weights[n + 1] = (uint8_t)(b & 15);
        }
    } else {
        if (iSize + 1 > srcSize) return kFail;
        nListed = dev_fse_weights(weights, 255, src + 1, iSize, scratch);
        if (nListed == kFail) return kFail;
    }
    return iSize + 1;
}

// Second half of HUF_readStats_body (:350-392) and the rank scan of HUF_readDTableX1_wksp (huf_decompress.pas:133-285)
// by a group of 16 lanes (a DPP row): weight counts, the implied last weight, and pos[n] = first table cell of
// symbol n (cells are laid out by weight, symbols in order inside a weight).  Lane l counts symbols 16 l .. 16 l + 15
// in four words of three 10-bit counters (weights 0..11), a prefix sum over the lanes gives every lane the counts
// before its symbols.  Returns false on an invalid description.  rank: 13 words of LDS.
DEVI bool grp_huf_ranks(uint8_t* weights, uint16_t* pos, uint32_t* rank, uint32_t nListed, uint32_t gl, uint32_t lead, uint32_t& nbSymbols, uint32_t& tableLog)
{
    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, wsum = 0;
    bool bad = false;
    uint32_t myW[16];
#pragma unroll
    for (int u = 0; u < 16; u++) {
        uint32_t const n = 16 * gl + u;
        uint32_t const w = n < nListed ? weights[n] : 0xFFu;
        myW[u] = w;
        if (n < nListed) {
            if (w >= 12) bad = true;
            uint32_t const inc = 1u << (10 * (w % 3));
            uint32_t const word = w / 3;
            c0 += word == 0 ? inc : 0u; c1 += word == 1 ? inc : 0u; c2 += word == 2 ? inc : 0u; c3 += word == 3 ? inc : 0u;
            wsum += (1u << (w & 15)) >> 1;
        }
    }
    if (((__ballot(bad) >> lead) & 0xFFFFu) != 0) return false;
    // inclusive prefix over the row
    uint32_t i0 = c0, i1 = c1, i2 = c2, i3 = c3, is = wsum;
#define ZG_ROW_STEP(CTRL) i0 += dpp_or0<CTRL, 0xF>(i0); i1 += dpp_or0<CTRL, 0xF>(i1); i2 += dpp_or0<CTRL, 0xF>(i2); i3 += dpp_or0<CTRL, 0xF>(i3); is += dpp_or0<CTRL, 0xF>(is);
    ZG_ROW_STEP(0x111) ZG_ROW_STEP(0x112) ZG_ROW_STEP(0x114) ZG_ROW_STEP(0x118)
#undef ZG_ROW_STEP
    uint32_t t0 = (uint32_t)__shfl((int)i0, (int)(lead + 15)), t1 = (uint32_t)__shfl((int)i1, (int)(lead + 15)),
             t2 = (uint32_t)__shfl((int)i2, (int)(lead + 15)), t3 = (uint32_t)__shfl((int)i3, (int)(lead + 15));
    uint32_t const weightTotal = (uint32_t)__shfl((int)is, (int)(lead + 15));
    if (weightTotal == 0) return false;
    tableLog = hb32(weightTotal) + 1;
    if (tableLog > 12) return false;
    uint32_t const rest = (1u << tableLog) - weightTotal;
    if ((1u << hb32(rest)) != rest) return false;
    uint32_t const lastWeight = hb32(rest) + 1;
    auto total_of = [&](uint32_t w) -> uint32_t { uint32_t const word = w / 3 == 0 ? t0 : w / 3 == 1 ? t1 : w / 3 == 2 ? t2 : t3; return (word >> (10 * (w % 3))) & 1023u; };
    uint32_t const lastBefore = total_of(lastWeight);          // symbols of that weight before the implied one
    {   // the implied symbol joins its class
        uint32_t const inc = 1u << (10 * (lastWeight % 3));
        uint32_t const word = lastWeight / 3;
        t0 += word == 0 ? inc : 0u; t1 += word == 1 ? inc : 0u; t2 += word == 2 ? inc : 0u; t3 += word == 3 ? inc : 0u;
    }
    uint32_t const n1 = total_of(1);
    if (n1 < 2 || (n1 & 1)) return false;
    nbSymbols = nListed + 1;
    // rank[w] := first cell of weight class w
    if (gl == 0) {
        uint32_t next = 0;
        rank[0] = 0;
        for (uint32_t w = 1; w <= 12; w++) { rank[w] = next; next += total_of(w) << (w - 1); }
        weights[nListed] = (uint8_t)lastWeight;
        pos[nListed] = (uint16_t)(rank[lastWeight] + (lastBefore << (lastWeight - 1)));
    }
    lds_sync();
    // this lane's symbols, on top of the counts of the lanes before it
    uint32_t r0 = i0 - c0, r1 = i1 - c1, r2 = i2 - c2, r3 = i3 - c3;
#pragma unroll
    for (int u = 0; u < 16; u++) {
        uint32_t const n = 16 * gl + u, w = myW[u];
        if (n < nListed) {
            uint32_t const word = w / 3 == 0 ? r0 : w / 3 == 1 ? r1 : w / 3 == 2 ? r2 : r3;
            uint32_t const before = (word >> (10 * (w % 3))) & 1023u;
            pos[n] = (uint16_t)(w ? rank[w] + (before << (w - 1)) : 0u);
            uint32_t const inc = 1u << (10 * (w % 3));
            r0 += w / 3 == 0 ? inc : 0u; r1 += w / 3 == 1 ? inc : 0u; r2 += w / 3 == 2 ? inc : 0u; r3 += w / 3 == 3 ? inc : 0u;
        }
    }
    lds_sync();
    return true;
}

// ---------------------------------------------------------------- stage 0: literals

constexpr int kHufLanes = 16;                // lanes per literal job: 4 decode one stream each, all 16 move data
constexpr int kHufJobs = 64 / kHufLanes;     // jobs per wave
constexpr uint32_t kHufTabBytes = 8192;      // 2^12 cells of {byte, nbBits}
constexpr uint32_t kHufRound = 128;          // symbols per stream per round
constexpr uint32_t kSpecParts = 8;          // split decode: parts per stream
constexpr uint32_t kPackTabEntries = 4 * kSpecParts + 1;    // per literal job: 32 parts + a flag
constexpr uint32_t kSpecRound = 32;         // split decode: symbols per lane per round
constexpr uint32_t kSpecWinDw = 24;         // split decode: staged input per lane (6 units): 32 symbols x <= 12 bits + alignment slack
constexpr uint32_t kHufInDw = 96;            // staged input per stream per round (24 units): 128 symbols x <=12 bits + alignment slack
constexpr uint32_t kHufStageBytes = 4 * 4 * kHufInDw + 4 * kHufRound;   // 4 windows + 4 output stages = 2048
constexpr uint32_t kHufJobBytes = kHufTabBytes + kHufStageBytes;        // the table-build scratch (weights/pos/rank, 820 B) shares the stage area

// Stream geometry of a literals section as scalars (arrays indexed by a lane-dependent stream number
// would live in scratch memory): stream k reads sOff(k)..+sSize(k) and writes oCnt(k) bytes at k * seg.
struct HufGeo {
    uint32_t l1, l2, l3, l4;   // compressed stream sizes
    uint32_t seg, lastCnt;     // bytes regenerated by streams 0..2 and by stream 3
    uint32_t first;            // offset of stream 0 (after the jump table)
    DEVI uint32_t sOff(uint32_t k) const { return first + (k > 0 ? l1 : 0u) + (k > 1 ? l2 : 0u) + (k > 2 ? l3 : 0u); }
    DEVI uint32_t sSize(uint32_t k) const { return k == 0 ? l1 : k == 1 ? l2 : k == 2 ? l3 : l4; }
    DEVI uint32_t oCnt(uint32_t k) const { return k < 3 ? seg : lastCnt; }
};

// HUF_selectDecoder and the two columns of its cost table that it reads (huf_decompress.pas:81-100, :1301-1318):
// true when the reference would decode (dstSize, cSrcSize) with the double-symbol decoder.  The bytes are the same
// either way; what differs is the verdict on a damaged stream at its last symbol (see the serial decode below).
__constant__ uint16_t c_algoTime[16][4] = {     // {tableTime, decode256Time} of the single-symbol and of the double-symbol decoder
    {0, 0, 1, 1}, {0, 0, 1, 1}, {38, 130, 1313, 74}, {448, 128, 1353, 74}, {556, 128, 1353, 74}, {714, 128, 1418, 74},
    {883, 128, 1437, 74}, {897, 128, 1515, 75}, {926, 128, 1613, 75}, {947, 128, 1729, 77}, {1107, 128, 2083, 81},
    {1177, 128, 2379, 87}, {1242, 128, 2415, 93}, {1349, 128, 2644, 106}, {1455, 128, 2422, 124}, {722, 128, 1891, 145}};
DEVI bool huf_select_x2(uint32_t dstSize, uint32_t cSrcSize)
{
    uint32_t const Q = cSrcSize >= dstSize ? 15u : cSrcSize * 16u / dstSize;
    uint32_t const D256 = dstSize >> 8;
    uint32_t const t0 = c_algoTime[Q][0] + c_algoTime[Q][1] * D256;
    uint32_t t1 = c_algoTime[Q][2] + c_algoTime[Q][3] * D256;
    t1 += t1 >> 3;
    return t1 < t0;
}

// flush up to kHufRound decoded bytes per stream from the LDS stages to the literal buffer: 32 units, 2 per lane
DEVI void huf_flush(uint8_t* out, const uint8_t* outStage, const HufGeo& g, uint32_t pr, uint32_t nStreams, uint32_t gl)
{
#pragma unroll
    for (int h = 0; h < 2; h++) {
        uint32_t const u = gl + 16 * h, k = u >> 3, part = u & 7;
        uint32_t const oc = g.oCnt(k);
        uint32_t const cnt = oc > pr ? (oc - pr < kHufRound ? oc - pr : kHufRound) : 0;
        if (k < nStreams && 16 * part < cnt) {
            uint8_t* const d = out + k * g.seg + pr + 16 * part;
            const uint8_t* const sp = outStage + k * kHufRound + 16 * part;
            if (16 * part + 16 <= cnt) { uint4 const x = *(const uint4*)sp; __builtin_memcpy(d, &x, 16); }
            else for (uint32_t i = 0; i < cnt - 16 * part; i++) d[i] = sp[i];
        }
    }
}

// kDictTable: every job of the launch is a single stream coded with the dictionary's table (small dictionary-compressed
// records).  One lane decodes such a stream, so the number of jobs a compute unit holds is what counts: the table is
// copied into LDS once per workgroup of 32 jobs (eight waves), not once per job, which lets a compute unit hold 64 jobs
// instead of 16.
constexpr int kHufDictWaves = 8;
constexpr int kHufDictJobs = kHufDictWaves * kHufJobs;
template <bool kDictTable>
__global__ __launch_bounds__(kDictTable ? 64 * kHufDictWaves : 64) __attribute__((amdgpu_waves_per_eu(kDictTable ? 4 : 1, kDictTable ? 4 : 2))) void k_literals(BatchParams p)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[kDictTable ? kHufTabBytes + kHufDictJobs * kHufStageBytes : kHufJobs * kHufJobBytes];
    uint32_t const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t const sub = lane / kHufLanes, sl = lane % kHufLanes;
    if (kDictTable) {
        // the dictionary's table, whole (its array holds 2^12 cells whatever its log), 16 bytes per thread
        static_assert(kHufTabBytes == 16 * 64 * kHufDictWaves, "one pass copies the table");
        ((uint4*)lds)[threadIdx.x] = ((const uint4*)p.dict->huf)[threadIdx.x];
        __syncthreads();
    }
    uint32_t const job = kDictTable ? blockIdx.x * kHufDictJobs + wave * kHufJobs + sub : blockIdx.x * kHufJobs + sub;
    if (job >= n_lit_jobs(p)) return;
    uint32_t const b = p.litJobs[job];
    DevBlock const blk = p.blocks[b];
    // (with the shared table a job's own area is the stage area alone: `tab` is placed so that the offsets below still hold)
    uint8_t* const tab = kDictTable ? lds + (wave * kHufJobs + sub) * kHufStageBytes : lds + sub * kHufJobBytes;
    uint16_t* const cells = kDictTable ? (uint16_t*)lds : (uint16_t*)tab;
    uint8_t* const weights = tab + kHufTabBytes;                       // table-build scratch, dead once the table is filled
    uint16_t* const pos = (uint16_t*)(tab + kHufTabBytes + 256);
    uint32_t* const rank = (uint32_t*)(tab + kHufTabBytes + 768);
    uint32_t* const inWin = (uint32_t*)(tab + kHufTabBytes);          // [4][kHufInDw] dwords
    uint8_t* const outStage = tab + kHufTabBytes + 4 * 4 * kHufInDw;  // [4][kHufRound] bytes

#ifdef ZG_STAMPS
    uint64_t const y0 = __builtin_readcyclecounter();
    uint64_t yA = 0, yB = 0, yC = 0, yD = 0, yN = 0, yS = 0;
#endif
    uint32_t err = 0, tableLog = 0, nbSymbols = 0, hSize = 0;
    bool const fromDict = (blk.hufSrc == SRC_DICT);
    // Which decoder the reference runs on this block's streams: the double-symbol one for a dictionary's table
    // (zstd_decompressf.pas:1288) and where HUF_selectDecoder picked it for the block that defined the table (four
    // streams only: a single-stream block always builds the single-symbol table, huf_decompress.pas:1386).
    bool x2 = fromDict && p.dict->hufX1 == 0;
    if (!kDictTable && blk.hufSrc >= 0) {
        uint32_t dSingle = blk.singleStream, dLit = blk.litSize, dC = blk.litCSize;
        if (blk.hufSrc != (int32_t)b) { const DevBlock* const sbp = p.blocks + blk.hufSrc; dSingle = sbp->singleStream; dLit = sbp->litSize; dC = sbp->litCSize; }
        x2 = !dSingle && dLit != 0 && huf_select_x2(dLit, dC);
    }
    // The table description (at most 129 bytes) is parsed by one lane; the group first copies it into LDS, with the
    // 8 bytes before it that the backward bit reader may touch (block and literals headers always precede it).
    uint8_t* const hdrStage = tab + 2048;
    uint32_t descAvail = 0;
    if (!kDictTable && blk.hufSrc >= 0) {
        DevBlock const sb = (blk.hufSrc == (int32_t)b) ? blk : p.blocks[blk.hufSrc];
        const uint8_t* const dp = p.src + sb.srcOff + sb.litHdrSize;
        descAvail = sb.litCSize < 136 ? sb.litCSize : 136;
        grp_stage_bytes(hdrStage, dp - 8, descAvail + 8, sl);
    }
    lds_sync();
    uint32_t nListed = 0;
    bool parse = false;         // a table description to digest (the group's first lane lists the weights, all 16 lanes rank them)
    if (sl == 0) {
        if (blk.litType == SET_COMPRESSED && !blk.singleStream && (blk.litSize == 0 || blk.litCSize == 0)) err = E_corruption;   // huf_decompress.pas:1421-1424
        else if (blk.hufSrc == SRC_NONE) err = E_corruption;
        else if (kDictTable && !fromDict) err = E_generic;      // (not launched on such a batch)
        else if (fromDict) tableLog = p.dict->hufLog;
        else {
            DevBlock const sb = (blk.hufSrc == (int32_t)b) ? blk : p.blocks[blk.hufSrc];
            uint32_t const r = dev_huf_listed_weights(hdrStage + 8, sb.litCSize, weights, tab, nListed);
            if (r == kFail) err = E_corruption;
            else {
                hSize = r;
                parse = true;
                if (blk.litType == SET_COMPRESSED && hSize >= blk.litCSize) err = E_corruption;   // huf_decompress.pas:1394, :1349
            }
        }
    }
    wave_sync();
#ifdef ZG_STAMPS
    uint64_t const yp = __builtin_readcyclecounter();
#endif
    uint32_t const lead = sub * kHufLanes;
    err = __shfl(err, lead); tableLog = __shfl(tableLog, lead); hSize = __shfl(hSize, lead); nListed = __shfl(nListed, lead);
    parse = __shfl((int)parse, lead) != 0;
    if (parse && !err) {
        if (!grp_huf_ranks(weights, pos, rank, nListed, sl, lead, nbSymbols, tableLog)) err = E_corruption;
    }

    bool ok = (err == 0);
    uint16_t* const x1 = cells;
    if (ok) {
        // table fill (HUF_readDTableX1_wksp second half): cell = {symbol, nbBits = tableLog + 1 - weight}
        if (kDictTable) {
        } else if (fromDict) {
            // 16 bytes per lane and pass, four passes' loads ahead of their stores (both arrays hold 4096 cells and are
            // 16-byte aligned: a short table is copied rounded up)
            uint32_t const units = ((2u << tableLog) + 15) >> 4;
            const uint4* const from = (const uint4*)p.dict->huf;
            uint4* const to = (uint4*)x1;
            for (uint32_t u = sl; u < units; u += 4 * kHufLanes) {
                uint32_t const u1 = u + kHufLanes < units ? u + kHufLanes : u, u2 = u + 2 * kHufLanes < units ? u + 2 * kHufLanes : u,
                               u3 = u + 3 * kHufLanes < units ? u + 3 * kHufLanes : u;
                uint4 const q0 = from[u], q1 = from[u1], q2 = from[u2], q3 = from[u3];
                to[u] = q0; to[u1] = q1; to[u2] = q2; to[u3] = q3;
            }
        } else {
            for (uint32_t n = sl; n < nbSymbols; n += kHufLanes) {
                uint32_t const w = weights[n];
                if (w) {
                    uint32_t const len = (1u << w) >> 1;
                    uint32_t const start = pos[n];
                    uint32_t const cell = n | ((tableLog + 1 - w) << 8);
                    fill_bytes((uint8_t*)(x1 + start), cell | (cell << 16), 2 * len);
                }
            }
        }
    }
    wave_sync();
#ifdef ZG_STAMPS
    uint64_t const yf = __builtin_readcyclecounter();
#endif
    wave_sync();

#ifdef ZG_STAMPS
    uint64_t const y1 = __builtin_readcyclecounter();
#endif
    // ---- stream geometry, known to every lane of the group (huf_decompress.pas:406-421)
    uint32_t const tabHdr = (blk.litType == SET_COMPRESSED) ? hSize : 0;
    const uint8_t* const cs = p.src + blk.srcOff + blk.litHdrSize + tabHdr;
    uint32_t const csize = blk.litCSize - tabHdr;
    uint8_t* const out = p.litBuf + blk.litBufOff;
    uint32_t const nStreams = blk.singleStream ? 1u : 4u;
    HufGeo g = {0, 0, 0, 0, 0, 0, 0};
    if (ok) {
        if (blk.singleStream) { g.l1 = csize; g.seg = blk.litSize; }
        else if (csize < 10) ok = false;
        else {
            g.l1 = cs[0] | (cs[1] << 8); g.l2 = cs[2] | (cs[3] << 8); g.l3 = cs[4] | (cs[5] << 8);
            g.seg = (blk.litSize + 3) / 4;
            g.first = 6;
            if (g.l1 + g.l2 + g.l3 + 6 > csize || 3 * g.seg > blk.litSize) ok = false;   // 4th segment must start inside the literal buffer
            else { g.l4 = csize - 6 - g.l1 - g.l2 - g.l3; g.lastCnt = blk.litSize - 3 * g.seg; }
        }
    }
    // ---- per-lane stream state (lanes 0..3 of the group own a stream)
    uint32_t const mySize = g.sSize(sl & 3), myCnt = g.oCnt(sl & 3);
    const uint8_t* const myStart = cs + g.sOff(sl & 3);
    int bitsLeft = 0;
    bool bad = !ok;
    if (ok && sl < nStreams) {
        // BIT_initDStream (bitstream.pas:206-275): empty stream or missing end mark is an error
        if (mySize == 0) bad = true;
        else {
            uint32_t const last = myStart[mySize - 1];
            if (last == 0) bad = true;
            else bitsLeft = (int)(mySize * 8) - (int)(8 - hb32(last));
        }
    }
    uint32_t const maxCnt = ok ? g.seg : 0;       // stream 0 is never shorter than the others
    uintptr_t const floorAddr = (uintptr_t)p.src;
    uint32_t const shift = 32 - tableLog;
    uint32_t done = 0;                            // symbols of my stream already decoded
    // ---- split decode: every stream in eight parts, two parts per lane.
    // A Huffman stream can only be decoded from its first code, but a decoder started at an arbitrary bit falls into
    // step with the true code boundaries after a few symbols.  Part P of stream k starts at bit P/8 of the stream and is
    // decoded into a scratch region of its own; a lane runs parts j and j + 4 of its stream side by side, so that the
    // table lookups of one fill the LDS round trips of the other.  At the start of each of its first 16 rounds (of 32
    // symbols) a part notes where it is.  A decoder that has reached the next part compares its own code boundaries with
    // those notes: at the first one it hits, the two decoders have met, the next part is on the true path from there on
    // (part 0 is from its start), and this decoder stops.  One that gets past all 16 notes without a hit carries on
    // through that part (whose own output is then dropped) towards the part after it.  A final copy packs the parts into
    // the literal buffer.  If the totals disagree with the stream's symbol count, or a scratch region overflows, the
    // block is decoded again stream by stream below (which also yields the verdict for corrupt streams).
    bool specDone = false;
    if (!kDictTable) {
        uint32_t const k = sl & 3, j = sl >> 2;
        int T = 0;
        bool sbad = !ok || nStreams != 4 || p.litTmp == nullptr || mySize == 0 || tableLog > 11;    // (a 2^12 table leaves no LDS for the windows)
        if (!sbad) {
            uint32_t const last = myStart[mySize - 1];
            if (last == 0) sbad = true;
            else T = (int)(mySize * 8) - (int)(8 - hb32(last));
        }
        if (T < (int)(kSpecParts * 1024)) sbad = true;       // (parts of at least 1024 bits)
        bool const eligible = ((__ballot(sbad) >> lead) & 0xFFFFu) == 0;
        if (eligible) {
            // LDS of the job beyond the 4 KB table: notes [32 parts][16], windows [32][kSpecWinDw], output stages [32][kSpecRound]
            uint16_t* const ckAll = (uint16_t*)(tab + 4096);
            uint32_t* const winAll = (uint32_t*)(tab + 4096 + 1024);
            uint8_t* const outAll = tab + 4096 + 1024 + 32 * 4 * kSpecWinDw;
            static_assert(4096 + 1024 + 32 * 4 * kSpecWinDw + 32 * kSpecRound <= kHufJobBytes, "split decode fits in the job's LDS");
            auto boundary = [&](uint32_t t) -> int { return T - (int)(((uint64_t)t * (uint32_t)T) >> 3); };     // first bit of part t (0 for t = 8)
            uint32_t const laneCap = litTmpPart(blk.litSize);
            uint8_t* const dump = p.litTmp + blk.litTmpOff + 32 * (size_t)laneCap;
            // per part of this lane (c = 0, 1; constant indices once unrolled)
            int cur[2], Bs[2], Bt[2];
            uint32_t t[2], nDec[2], upTo[2], hdT[2], q[2], ckv[2], rem[2];
            bool met[2], failed[2], prevActive[2];
            uintptr_t wbL[2], pwb[2];
            // The windows requested during the previous round.  A request is an address in LDS; the 32 x 6 units of a block are
            // loaded by its 16 lanes twelve apiece, unit after unit, so that the six units of a window (96 contiguous bytes)
            // sit in neighbouring lanes of one load and coalesce -- a lane loading its own windows would put 64 unrelated
            // addresses into every load.
            uint4 pvT[12];
            uint32_t pvMask = 0;            // which of them hold a requested unit
            uint64_t* const reqAll = (uint64_t*)(tab + 4096 + 1024 + 32 * 4 * kSpecWinDw + 32 * kSpecRound);
            static_assert(4096 + 1024 + 32 * 4 * kSpecWinDw + 32 * kSpecRound + 32 * 8 <= kHufJobBytes, "window requests fit in the job's LDS");
            uint8_t* tmpL[2]; uint32_t* winL[2]; uint32_t* soL[2]; uint16_t* ckL[2];
#pragma unroll
            for (int c = 0; c < 2; c++) {
                uint32_t const reg = sl + 16 * c, P = j + 4 * c;
                Bs[c] = boundary(P); t[c] = P + 1; Bt[c] = boundary(P + 1); cur[c] = Bs[c];
                nDec[c] = upTo[c] = hdT[c] = q[c] = rem[c] = 0; ckv[c] = 0xFFFFu;
                met[c] = failed[c] = prevActive[c] = false;
                wbL[c] = pwb[c] = 0;
                tmpL[c] = p.litTmp + blk.litTmpOff + (size_t)reg * laneCap;
                winL[c] = winAll + reg * kSpecWinDw;
                soL[c] = (uint32_t*)(outAll + reg * kSpecRound);
                ckL[c] = ckAll + 16 * reg;
                *(uint4*)ckL[c] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);      // "not noted yet": matches nothing
                *(uint4*)(ckL[c] + 8) = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
            }
            bool hasPv[2] = {false, false};
            int lastCur[2] = {cur[0], cur[1]};
            uint32_t lastUse[2] = {tableLog * kSpecRound, tableLog * kSpecRound};      // bits the previous round consumed
#ifdef ZG_STAMPS
            uint64_t zA = 0, zB = 0, zC = 0, zR = 0, zF = 0; uint64_t const z0 = __builtin_readcyclecounter();
#endif
            for (uint32_t r = 0;; r++) {
#ifdef ZG_STAMPS
                uint64_t const za = __builtin_readcyclecounter();
#endif
                bool act[2];
#pragma unroll
                for (int c = 0; c < 2; c++) act[c] = !met[c] && !failed[c];
                bool const anyActive = ((__ballot(act[0] || act[1]) >> lead) & 0xFFFFu) != 0;
                // where each part is at the start of this round
                if (r < 16) {
#pragma unroll
                    for (int c = 0; c < 2; c++) { uint32_t const d = (uint32_t)(Bs[c] - cur[c]); ckL[c][r] = (uint16_t)(d < 0xFFFEu ? d : 0xFFFEu); }
                }
                // Windows: 6 units per part ending at the byte that holds its cursor.  Every round requests the windows of the
                // next one, so the load is over when it is needed; only when a round eats so much that such a window cannot feed
                // the next round is one loaded on the spot.  (Byte-aligned loads lose at most 7 bits to alignment; a window that
                // would reach below the source buffer keeps to 16-byte units, which the clamp in the load needs.)
                auto window_base = [&](int cr) -> uintptr_t {
                    uintptr_t const top = (uintptr_t)myStart + (uintptr_t)((cr > 0 ? cr + 7 : 0) >> 3);
                    uintptr_t const exact = top - 4 * kSpecWinDw;
                    return exact >= floorAddr ? exact : ((top + 15) & ~(uintptr_t)15) - 4 * kSpecWinDw;
                };
                bool low = false;
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    lastUse[c] = r ? (uint32_t)(lastCur[c] - cur[c]) : lastUse[c]; lastCur[c] = cur[c];
                    uintptr_t const wbCand = hasPv[c] ? pwb[c] : wbL[c];
                    low = low || (act[c] && (r == 0 || (int)(8 * (int64_t)((uintptr_t)myStart - wbCand)) + cur[c] < (int)(tableLog * kSpecRound + 96)));
                }
                bool const stage = anyActive && ((__ballot(low) >> lead) & 0xFFFFu) != 0;
                uint4 v[2][6];
                uintptr_t wbS[2] = {0, 0};
                if (stage) {
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        wbS[c] = window_base(cur[c]);
#pragma unroll
                        for (int u = 0; u < 6; u++) v[c][u] = ld_unit_any(wbS[c] + 16 * u, floorAddr);
                    }
                }
                // (every lane stores, a part with nothing to flush into the spare bytes behind the block's regions: a fixed
                // number of stores lets the wait for the window loads issued ahead of them be counted)
                auto flush_prev = [&]() {
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        uint4 const x0 = *(const uint4*)soL[c], x1v = *(const uint4*)(soL[c] + 4);
                        uint8_t* const d = prevActive[c] ? tmpL[c] + kSpecRound * (r - 1) : dump;
                        __builtin_memcpy(d, &x0, 16);
                        __builtin_memcpy(d + 16, &x1v, 16);
                    }
                };
                if (!anyActive) { flush_prev(); break; }
#ifdef ZG_STAMPS
                uint64_t const zb = __builtin_readcyclecounter();
#endif
                lds_sync();
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    if (stage) {
                        wbL[c] = wbS[c];
#pragma unroll
                        for (int u = 0; u < 6; u++) *(uint4*)(winL[c] + 4 * u) = v[c][u];
                    } else if (hasPv[c]) wbL[c] = pwb[c];
                    hasPv[c] = false;
                }
                if (!stage) {
#pragma unroll
                    for (uint32_t i = 0; i < 12; i++) {
                        uint32_t const gi = 16 * i + sl, X = (gi * 171u) >> 10, u = gi - 6 * X;      // unit u of chain X (gi / 6 for gi < 192)
                        if ((pvMask >> i) & 1) *(uint4*)(winAll + X * kSpecWinDw + 4 * u) = pvT[i];
                    }
                }
                pvMask = 0;
                lds_sync();
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    // The window the next round will start from -- when the one in place would not feed it: what is left now,
                    // less what this round will take (about what the last one took), against the most a round can take.
                    int const left = (int)(8 * (int64_t)((uintptr_t)myStart - wbL[c])) + cur[c];
                    bool const need = act[c] && left - (int)(lastUse[c] + lastUse[c] / 4 + 16) < (int)(tableLog * kSpecRound + 96);
                    if (need) { pwb[c] = window_base(cur[c]); hasPv[c] = true; }
                    reqAll[sl + 16 * c] = need ? (uint64_t)pwb[c] : 0ull;
                }
                lds_sync();
#pragma unroll
                for (uint32_t i = 0; i < 12; i++) {
                    uint32_t const gi = 16 * i + sl, X = (gi * 171u) >> 10, u = gi - 6 * X;
                    uint64_t const base = reqAll[X];
                    pvT[i] = ld_unit_any(base ? (uintptr_t)base + 16 * u : floorAddr, floorAddr);       // (every lane loads: a fixed number of loads in flight)
                    pvMask |= base ? (1u << i) : 0u;
                }
                // The previous round's bytes go to the parts' scratch regions: behind the window requests, which must not wait
                // for these stores to be acknowledged.
                flush_prev();
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    prevActive[c] = false;
                    if (act[c]) {
                        if (kSpecRound * (r + 1) > laneCap) { failed[c] = true; act[c] = false; }
                        else prevActive[c] = true;
                    }
                }
#ifdef ZG_STAMPS
                uint64_t const zc = __builtin_readcyclecounter();
                zA += zb - za; zB += zc - zb; zR += 1;
#endif
                if (act[0] || act[1]) {
                    int Qbase[2];
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        Qbase[c] = (int)(8 * (int64_t)((uintptr_t)myStart - wbL[c]));
                        if (r == 0) { uint32_t lo; peek64(winL[c], (uint32_t)(Qbase[c] + cur[c]), rem[c], lo); }
                    }
                    // Always the whole round: a part that has met the one ahead decodes on harmlessly inside its window; what it
                    // produces past that point is never used.  A part that was not active when the round began stays where it is.
                    // Most rounds no part of the wave can reach the part ahead of it (or the end of its stream): those rounds run
                    // without any of the meeting logic.
                    bool nearTarget = false;
#pragma unroll
                    for (int c = 0; c < 2; c++) nearTarget = nearTarget || (act[c] && cur[c] - (int)(tableLog * kSpecRound + 64) <= Bt[c]);
#ifdef ZG_STAMPS
                    zF += __any(nearTarget) ? 0 : 1;
#endif
                    if (!__any(nearTarget)) {
#pragma unroll
                        for (uint32_t gq = 0; gq < kSpecRound / 4; gq++) {
                            uint32_t tHi[2], tLo[2], c0[2], c1[2], c2[2], c3[2], l0[2], l1[2], l2[2], l3[2];
#pragma unroll
                            for (int c = 0; c < 2; c++) peek64(winL[c], (uint32_t)(Qbase[c] + cur[c]), tHi[c], tLo[c]);
#pragma unroll
                            for (int c = 0; c < 2; c++) c0[c] = cells[rem[c] >> shift];
#pragma unroll
                            for (int c = 0; c < 2; c++) { l0[c] = c0[c] >> 8; c1[c] = cells[shl_hi(tHi[c], tLo[c], l0[c]) >> shift]; }
#pragma unroll
                            for (int c = 0; c < 2; c++) { l1[c] = l0[c] + (c1[c] >> 8); c2[c] = cells[shl_hi(tHi[c], tLo[c], l1[c]) >> shift]; }
#pragma unroll
                            for (int c = 0; c < 2; c++) { l2[c] = l1[c] + (c2[c] >> 8); c3[c] = cells[shl_hi(tHi[c], tLo[c], l2[c]) >> shift]; }
#pragma unroll
                            for (int c = 0; c < 2; c++) {
                                l3[c] = l2[c] + (c3[c] >> 8);
                                soL[c][gq] = (c0[c] & 0xFF) | ((c1[c] & 0xFF) << 8) | ((c2[c] & 0xFF) << 16) | (c3[c] << 24);
                                uint32_t const adv = act[c] ? l3[c] : 0u;
                                cur[c] -= (int)adv;
                                rem[c] = shl_hi(tHi[c], tLo[c], adv);
                            }
                        }
#pragma unroll
                        for (int c = 0; c < 2; c++) nDec[c] += act[c] ? kSpecRound : 0u;
                    } else
#pragma unroll
                    for (uint32_t gq = 0; gq < kSpecRound / 4; gq++) {
                        uint32_t tHi[2], tLo[2], c0[2], c1[2], c2[2], c3[2], l0[2], l1[2], l2[2], l3[2];
                        // the two parts' lookups alternate: each waits for its own previous one while the other's is in flight
#pragma unroll
                        for (int c = 0; c < 2; c++) peek64(winL[c], (uint32_t)(Qbase[c] + cur[c]), tHi[c], tLo[c]);
                        // (the first code of a group is looked up in what the previous group left of its 64 bits -- at least 20 --
                        // while the window words of this group are on their way)
#pragma unroll
                        for (int c = 0; c < 2; c++) c0[c] = cells[rem[c] >> shift];
#pragma unroll
                        for (int c = 0; c < 2; c++) { l0[c] = c0[c] >> 8; c1[c] = cells[shl_hi(tHi[c], tLo[c], l0[c]) >> shift]; }
#pragma unroll
                        for (int c = 0; c < 2; c++) { l1[c] = l0[c] + (c1[c] >> 8); c2[c] = cells[shl_hi(tHi[c], tLo[c], l1[c]) >> shift]; }
#pragma unroll
                        for (int c = 0; c < 2; c++) { l2[c] = l1[c] + (c2[c] >> 8); c3[c] = cells[shl_hi(tHi[c], tLo[c], l2[c]) >> shift]; }
#pragma unroll
                        for (int c = 0; c < 2; c++) {
                            l3[c] = l2[c] + (c3[c] >> 8);
                            soL[c][gq] = (c0[c] & 0xFF) | ((c1[c] & 0xFF) << 8) | ((c2[c] & 0xFF) << 16) | (c3[c] << 24);
                        }
#pragma unroll
                        for (int c = 0; c < 2; c++) {
                            // code boundaries of the four symbols
                            int const cb0 = cur[c], cb1 = cur[c] - (int)l0[c], cb2 = cur[c] - (int)l1[c], cb3 = cur[c] - (int)l2[c];
                            bool const open0 = act[c] && !met[c] && !failed[c];
                            if (__any(open0 && cur[c] - 48 <= Bt[c])) {
                                // in the part ahead: is one of the boundaries where that part noted to be?  (the end of the stream: is
                                // one of them bit 0?)
                                bool const last = t[c] == kSpecParts;
                                ckv[c] = (!last && q[c] < 16) ? (uint32_t)ckAll[16 * (k + 4 * (last ? 0u : t[c])) + (q[c] < 16 ? q[c] : 0u)] : 0xFFFFu;
                                bool const valid = last || ckv[c] < 0xFFFEu;      // (0xFFFF: not noted yet, or no notes left)
                                int const goal = last ? 0 : Bt[c] - (int)ckv[c];
                                bool const m0 = valid && cb0 == goal, m1 = valid && cb1 == goal, m2 = valid && cb2 == goal, m3 = valid && cb3 == goal;
                                if ((m0 | m1 | m2 | m3) && open0) {
                                    met[c] = true;
                                    upTo[c] = nDec[c] + (m0 ? 0u : m1 ? 1u : m2 ? 2u : 3u);
                                    hdT[c] = kSpecRound * q[c];
                                }
                            }
                            if (act[c]) {
                                nDec[c] += 4;
                                cur[c] -= (int)l3[c];
                                rem[c] = shl_hi(tHi[c], tLo[c], l3[c]);
                                bool const open = !met[c] && !failed[c];
                                bool const atEnd = open && t[c] == kSpecParts;
                                if (atEnd && cur[c] == 0) { met[c] = true; upTo[c] = nDec[c]; }
                                if (atEnd && cur[c] < 0) failed[c] = true;
                                if (open && t[c] < kSpecParts) {
                                    // past the note in hand: on to the next one; past the last one without a hit: on through that part
                                    if (ckv[c] < 0xFFFEu && Bt[c] - cur[c] > (int)ckv[c]) { q[c]++; ckv[c] = 0xFFFFu; }
                                    if (q[c] >= 16) { t[c]++; Bt[c] = boundary(t[c]); q[c] = 0; ckv[c] = 0xFFFFu; }
                                }
                            }
                        }
                    }
                }
                lds_sync();
#ifdef ZG_STAMPS
                zC += __builtin_readcyclecounter() - zc;
#endif
            }
#ifdef ZG_STAMPS
            if (p.dbg && sl == 0) { uint64_t* const qd = p.dbg + 16 * (size_t)p.nBlocks + 16 * (size_t)b; qd[1] = zA; qd[2] = zB; qd[3] = zC; qd[4] = __builtin_readcyclecounter() - z0; qd[5] = zR; qd[6] = zF; }
#endif
            // ---- what every part found: {target part, symbols before the meeting point, symbols the target decoded before it, ok}
            lds_sync();
            uint4* const recs = (uint4*)outAll;       // [32 parts], the output stages are free now
#pragma unroll
            for (int c = 0; c < 2; c++) recs[sl + 16 * c] = make_uint4(t[c], upTo[c], hdT[c], (met[c] && !failed[c]) ? 1u : 0u);
            lds_sync();
            // walk the stream's parts along the true path: part 0, then the part it met, and so on to the end
            uint32_t val[2] = {0, 0}, hd[2] = {0, 0}, dOff[2] = {0, 0};
            bool chainOk = true;
            uint32_t acc = 0;
            {
                uint32_t part = 0, head = 0;
#pragma unroll
                for (int step = 0; step < (int)kSpecParts; step++) {
                    if (part < kSpecParts) {
                        uint4 const rc = recs[k + 4 * part];
                        uint32_t const vv = rc.y - head;
                        if (!rc.w || rc.y < head || rc.y > laneCap) chainOk = false;
#pragma unroll
                        for (int c = 0; c < 2; c++) {
                            uint32_t const P = j + 4 * c;
                            if (P == part) { val[c] = vv; hd[c] = head; dOff[c] = acc; }
                            if (P > part && P < rc.x) { val[c] = 0; hd[c] = 0; dOff[c] = acc + vv; }      // a part the path skips
                        }
                        acc += vv;
                        head = rc.z;
                        part = rc.x;
                    }
                }
                if (part != kSpecParts) chainOk = false;
            }
            bool const lbad = !chainOk || acc != myCnt;
            specDone = ((__ballot(lbad) >> lead) & 0xFFFFu) == 0;
#ifdef ZG_STAMPS
            if (p.dbg) {
                uint64_t* const qd = p.dbg + 16 * (size_t)p.nBlocks + 16 * (size_t)b;
                atomicOr((unsigned long long*)&qd[12], (unsigned long long)(64u | ((failed[0] || failed[1]) ? 1u : 0u) | (!chainOk ? 2u : 0u) | (lbad ? 4u : 0u) | ((t[0] != j + 1 || t[1] != j + 5) ? 8u : 0u)));
            }
#endif
            if (specDone) {
                // ---- leave the part table for the pack kernel, in output order: {first byte, end} in the literal buffer and the
                // address that byte 0 of the buffer would have in the part's scratch region
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    uint32_t const P = j + 4 * c;
                    uint32_t const dStart = k * g.seg + dOff[c];
                    uintptr_t const delta = (uintptr_t)(tmpL[c] + hd[c]) - dStart;
                    p.packTab[(size_t)job * kPackTabEntries + kSpecParts * k + P] = make_uint4(dStart, dStart + val[c], (uint32_t)delta, (uint32_t)((uint64_t)delta >> 32));
                }
            }
        }
    }
    uint32_t const maxCntSerial = specDone ? 0u : maxCnt;
    uint32_t pend = 0;      // serial decode under the double-symbol decoder: see cell_step below

    // Input windows: per stream 24 aligned 16-byte units ending just above its unread bits.  A round consumes at most
    // 1536 bits (usually far less); the windows are loaded again only when some stream's window could not feed
    // another round.
    uintptr_t wb[4] = {0, 0, 0, 0};
    for (uint32_t r0 = 0; r0 < maxCntSerial; r0 += kHufRound) {
#ifdef ZG_STAMPS
        uint64_t const ya = __builtin_readcyclecounter();
#endif
        // 1. windows: does any stream of this block need a new one?
        uintptr_t const curWb = sl == 0 ? wb[0] : sl == 1 ? wb[1] : sl == 2 ? wb[2] : wb[3];
        bool const low = sl < nStreams && !bad && done < myCnt &&
                         (int)(8 * (int64_t)((uintptr_t)myStart - curWb)) + bitsLeft < (int)(12 * kHufRound + 96);
        bool const stage = r0 == 0 || ((__ballot(low) >> lead) & 0xFFFFu) != 0;
        uint4 v[6];
        if (stage) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                int const bl = __shfl(bitsLeft, lead + k);
                uintptr_t const top = (uintptr_t)(cs + g.sOff(k)) + (uintptr_t)((bl > 0 ? bl + 7 : 0) >> 3);
                wb[k] = ((top + 15) & ~(uintptr_t)15) - 4 * kHufInDw;
            }
#pragma unroll
            for (int j = 0; j < 6; j++) {
                uint32_t const u = sl + 16 * j, k = (u * 43u) >> 10, iu = u - 24 * k;    // u / 24 for u < 96
                uintptr_t const base = k == 0 ? wb[0] : k == 1 ? wb[1] : k == 2 ? wb[2] : wb[3];
                v[j] = ld_unit(base + 16 * iu, floorAddr);
            }
        }
#ifdef ZG_STAMPS
        uint64_t const yb = __builtin_readcyclecounter();
#endif
        // 2. flush the previous round's bytes (stores younger than the loads above: not waited for in this round)
        if (r0 > 0) huf_flush(out, outStage, g, r0 - kHufRound, nStreams, sl);
        lds_sync();     // the flush has read the stages
#ifdef ZG_STAMPS
        uint64_t const yc = __builtin_readcyclecounter();
#endif
        if (stage) {
#pragma unroll
            for (int j = 0; j < 6; j++) *(uint4*)(inWin + 4 * (sl + 16 * j)) = v[j];
        }
        lds_sync();
#ifdef ZG_STAMPS
        uint64_t const yd = __builtin_readcyclecounter();
#endif
        // 3. decode up to 128 symbols of my stream, LDS only (HUF_decodeStreamX1, huf_decompress.pas:301-349)
        if (sl < nStreams && !bad && done < myCnt) {
            // (under the double-symbol decoder the stream's last symbol has rules of its own: it is left to the end of the step)
            uint32_t const myMain = x2 ? myCnt - 1 : myCnt;
            uint32_t const n = myMain - done < kHufRound ? myMain - done : kHufRound;
            uintptr_t const mywb = sl == 0 ? wb[0] : sl == 1 ? wb[1] : sl == 2 ? wb[2] : wb[3];
            int const Qbase = (int)(8 * (int64_t)((uintptr_t)myStart - mywb));
            uint32_t Q = (uint32_t)(Qbase + bitsLeft);       // >= 1409: a round consumes <= 1536 bits of a window that holds >= 2945
            const uint32_t* const win = inWin + sl * kHufInDw;
            uint32_t* const so = (uint32_t*)(outStage + sl * kHufRound);
            // The double-symbol decoder takes the symbols in cells of one or two: two when the second code fits in the 12 bits
            // behind the first (HUF_fillDTableX2, huf_decompress.pas:741-790).  `pend` = length of the code that opened the
            // cell in progress (0: the next symbol opens one); it decides how the stream's last symbol is judged.
            auto cell_step = [&](uint32_t len) { bool const second = pend != 0 && pend + len <= 12; pend = second ? 0u : len; };
            uint32_t i = 0;
            for (; i + 4 <= n; i += 4) {     // 4 symbols (<= 48 bits) per 64-bit peek
                uint32_t tHi, tLo;
                peek64(win, Q, tHi, tLo);
                uint32_t const c0 = cells[tHi >> shift];
                uint32_t pbits = c0 >> 8;
                uint32_t const c1 = cells[shl_hi(tHi, tLo, pbits) >> shift];
                pbits += c1 >> 8;
                uint32_t const c2 = cells[shl_hi(tHi, tLo, pbits) >> shift];
                pbits += c2 >> 8;
                uint32_t const c3 = cells[shl_hi(tHi, tLo, pbits) >> shift];
                pbits += c3 >> 8;
                so[i >> 2] = (c0 & 0xFF) | ((c1 & 0xFF) << 8) | ((c2 & 0xFF) << 16) | (c3 << 24);
                Q -= pbits;
                if (x2) { cell_step(c0 >> 8); cell_step(c1 >> 8); cell_step(c2 >> 8); cell_step(c3 >> 8); }
            }
            for (; i < n; i++) {
                uint32_t tHi, tLo;
                peek64(win, Q, tHi, tLo);
                uint32_t const c0 = cells[tHi >> shift];
                ((uint8_t*)so)[i] = (uint8_t)c0;
                Q -= c0 >> 8;
                if (x2) cell_step(c0 >> 8);
            }
            done += n;
            if (x2 && done == myMain && n < kHufRound) {
                // The stream's last symbol under the double-symbol decoder, worked out on the single-symbol table.  r = bits of
                // the stream not yet read; look-ups past them see zeros.
                //  - It closes a cell of two (the cell was looked up when its first code was read): an ordinary symbol, the
                //    stream must end with its code.
                //  - It opens a cell: HUF_decodeLastSymbolX2 (huf_decompress.pas:923-942) writes the cell's first symbol; a
                //    cell of one consumes its code as usual; a cell of two consumes both codes unless the stream is used
                //    up already, and a count that then passes the end of the stream is set back to it -- such a stream ends
                //    well whatever the codes were.  With nothing left (r = 0) the look-up's shift count wraps
                //    (bitstream.pas:310-314) and the 12 bits are the top of the container in hand: the stream's first 8 bytes.
                int const r = (int)Q - Qbase;
                uint32_t tHi, tLo;
                peek64(win, Q, tHi, tLo);
                uint32_t wz = tHi >> 20;
                if (r < 12) wz &= r > 0 ? ~((1u << (12 - r)) - 1u) : 0u;
                uint32_t const cZ = cells[wz >> (12 - tableLog)], nZ = cZ >> 8;
                bool fine;
                uint32_t byte;
                if (pend != 0 && pend + nZ <= 12) { byte = cZ; fine = (int)nZ == r; }
                else {
                    uint32_t w12 = wz;
                    if (r <= 0) {
                        uint64_t c8 = 0;
                        for (uint32_t q = 0; q < 8 && q < mySize; q++) c8 |= (uint64_t)myStart[q] << (8 * q);
                        w12 = (uint32_t)(c8 >> 52);
                    }
                    uint32_t const cA = cells[w12 >> (12 - tableLog)], n1 = cA >> 8;
                    uint32_t const cB = cells[((w12 << n1) & 0xFFFu) >> (12 - tableLog)], n2 = cB >> 8;
                    bool const two = n1 + n2 <= 12;
                    fine = r >= 0 && (two ? (r == 0 || (int)(n1 + n2) >= r) : (int)n1 == r);
                    byte = cA;
                }
                ((uint8_t*)so)[n] = (uint8_t)byte;
                if (fine) Q = (uint32_t)Qbase; else bad = true;
                done += 1;
            }
            bitsLeft = (int)Q - Qbase;
        }
        lds_sync();
#ifdef ZG_STAMPS
        { uint64_t const ye = __builtin_readcyclecounter(); yA += yb - ya; yB += yc - yb; yC += yd - yc; yD += ye - yd; yN += 1; yS += stage ? 1 : 0; }
#endif
    }
    if (maxCntSerial > 0) huf_flush(out, outStage, g, ((maxCntSerial - 1) / kHufRound) * kHufRound, nStreams, sl);
    // every stream must end exactly at its first bit (BIT_endOfDStream, huf_decompress.pas:553-555)
    if (sl < nStreams && !bad && !specDone && bitsLeft != 0) bad = true;
    unsigned long long const anyBad = __ballot(bad);
    if (sl == 0) {
        uint32_t const mine = (uint32_t)((anyBad >> lead) & ((1u << kHufLanes) - 1));
        p.state[b].litErr = err ? err : (mine ? (uint32_t)E_corruption : 0u);
        p.packTab[(size_t)job * kPackTabEntries + 4 * kSpecParts] = make_uint4(specDone ? 1u : 0u, 0, 0, 0);     // does the pack kernel have work for this block?
#ifdef ZG_STAMPS
        if (p.dbg) { uint64_t* const q = p.dbg + 16 * (size_t)p.nBlocks + 16 * (size_t)b; q[0] = y1 - y0; if (!specDone) { q[1] = yA; q[2] = yB; q[3] = yC; q[4] = yD; q[5] = yN; q[6] = yS; } q[7] = __builtin_readcyclecounter() - y0; q[8] = blk.litSize; q[9] = yp - y0; q[10] = yf - yp; q[11] = 0; }
#endif
    }
}

// Literals of a block that were decoded in 32 parts (split decode above): copy every part's valid bytes from its
// scratch region to its place in the literal buffer.  One workgroup per block; the destination is dense, so the
// workgroup walks it in 16-byte units (all loads of a thread first, then its stores); a unit that lies in one part is
// one 16-byte copy, a unit that spans parts goes byte by byte.
typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(1)));     // 16 bytes at any address
constexpr int kPackThreads = 256;
constexpr int kPackUnits = 12;      // units per thread and pass: 256 x 12 x 16 bytes = 48 KB, a whole block's literals in most cases
__global__ __launch_bounds__(kPackThreads) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_pack_literals(BatchParams p)
{
    __shared__ uint4 parts[32];
    uint32_t const job = blockIdx.x, tid = threadIdx.x;
    if (job >= n_lit_jobs(p)) return;
    const uint4* const tab = p.packTab + (size_t)job * kPackTabEntries;
    if (tab[32].x == 0) return;
    if (tid < 32) parts[tid] = tab[tid];
    __syncthreads();
    DevBlock const blk = p.blocks[p.litJobs[job]];
    uint8_t* const out = p.litBuf + blk.litBufOff;
    uint32_t const litSize = blk.litSize, nUnits = (litSize + 15) >> 4;
    // part of buffer byte d: the first one that ends after d
    auto part_of = [&](uint32_t d) -> uint32_t {
        uint32_t q = 0;
#pragma unroll
        for (int stepq = 16; stepq >= 1; stepq >>= 1) q += parts[q + stepq - 1].y <= d ? (uint32_t)stepq : 0u;
        return q < 31 ? q : 31u;
    };
    for (uint32_t first = 0; first < nUnits; first += kPackThreads * kPackUnits) {
        uint4 x[kPackUnits];
        bool whole[kPackUnits];
#pragma unroll
        for (int e = 0; e < kPackUnits; e++) {
            uint32_t const u = first + (uint32_t)e * kPackThreads + tid, d = 16 * u;
            whole[e] = false;
            if (u < nUnits) {
                uint4 const pt = parts[part_of(d)];
                whole[e] = d >= pt.x && d + 16 <= pt.y;
                // (a unit that is not copied whole reads the start of the scratch buffer instead: the loads stay unconditional)
                const uint8_t* const from = whole[e] ? (const uint8_t*)((((uintptr_t)pt.w) << 32 | pt.z) + d) : p.litTmp;
                u32x4u const val = *(const u32x4u*)from;
                x[e] = make_uint4(val.x, val.y, val.z, val.w);
            }
        }
#pragma unroll
        for (int e = 0; e < kPackUnits; e++) {
            uint32_t const u = first + (uint32_t)e * kPackThreads + tid;
            if (whole[e]) *(uint4*)(out + 16 * (size_t)u) = x[e];
        }
    }
    // units that span parts (or the end of the buffer) each hold the end of some part: one thread per part gathers
    // the bytes of that unit, all loads in flight together
    if (tid < 32) {
        uint32_t const e = parts[tid].y;
        if (e & 15u) {
            uint32_t const d = e & ~15u;
            uint8_t by[16];
#pragma unroll
            for (int c = 0; c < 16; c++) {
                uint32_t const bq = d + (uint32_t)c < litSize ? d + (uint32_t)c : litSize - 1;
                uint4 const pc = parts[part_of(bq)];
                by[c] = *(const uint8_t*)((((uintptr_t)pc.w) << 32 | pc.z) + bq);
            }
#pragma unroll
            for (int c = 0; c < 16; c++) if (d + (uint32_t)c < litSize) out[d + c] = by[c];
        }
    }
}

// ---------------------------------------------------------------- sequence tables
// Packed decoding cell of a sequence table, laid out so that the SUM of the three cells of a sequence
// (literal length, match length, offset) carries the bit counts the cursor needs:
//   bits  0-8   nextState base
//   bits  9-13  symbol-code bits (LL: code mod 16, ML: code mod 32, OF: none)
//   bits 14-18  (32 - nbBits) mod 32: summed over the cells of the tables read before and this one, mod 32, it is where
//               this table's bits start in the 32 bits that follow the additional bits
//   bits 21-25  nbBits (<= 9; the low 19 bits of three cells sum to less than 2^21, the sum of three here is <= 26)
//   bits 26-31  16 (LL, ML) or 31 (OF) - nbAdditionalBits: the sum of three is 63 - (additional bits of the sequence)
// The code bits hold code mod 16 (LL) or code mod 32 (ML); together with nbAdditionalBits they identify the
// code (LL codes 0-15 and ML codes 0-31 have no additional bits, the higher ones are told apart by theirs).
// An offset code equals its nbAdditionalBits.  The base value of a code comes from a small LDS side table.
enum { T_LL = 0, T_OF = 1, T_ML = 2 };
DEVI uint32_t seq_cell(int t, uint32_t nextState, uint32_t nbBits, uint32_t addBits, uint32_t code)
{
    uint32_t const low = t == T_LL ? (code & 15u) : t == T_ML ? (code & 31u) : 0u;
    return nextState | (low << 9) | (((32u - nbBits) & 31u) << 14) | (nbBits << 21) | (((t == T_OF ? 31u : 16u) - addBits) << 26);
}
DEVI uint32_t cell_addBits(int t, uint32_t c) { return (t == T_OF ? 31u : 16u) - (c >> 26); }
DEVI uint32_t cell_nbBits(uint32_t c) { return __builtin_amdgcn_ubfe(c, 21, 4); }
DEVI uint32_t cell_llCode(uint32_t c)
{
    uint32_t const low = (c >> 9) & 15u, a = cell_addBits(T_LL, c);
    return low + (a == 0 ? 0u : a >= 13 ? 32u : 16u);
}
DEVI uint32_t cell_mlCode(uint32_t c)
{
    uint32_t const low = (c >> 9) & 31u;
    return low + (cell_addBits(T_ML, c) ? 32u : 0u);
}

__constant__ uint32_t c_LL_base[36] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40,
                                       48, 64, 0x80, 0x100, 0x200, 0x400, 0x800, 0x1000, 0x2000, 0x4000, 0x8000, 0x10000};
__constant__ uint8_t c_LL_bits[36] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3,
                                      4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
__constant__ uint32_t c_ML_base[53] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26,
                                       27, 28, 29, 30, 31, 32, 33, 34, 35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 0x83, 0x103,
                                       0x203, 0x403, 0x803, 0x1003, 0x2003, 0x4003, 0x8003, 0x10003};
__constant__ uint8_t c_ML_bits[53] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                      0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11,
                                      12, 13, 14, 15, 16};
__constant__ int16_t c_LL_defaultNorm[36] = {4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2,
                                             2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
__constant__ int16_t c_ML_defaultNorm[53] = {1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                             1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1,
                                             -1, -1, -1, -1, -1};
__constant__ int16_t c_OF_defaultNorm[29] = {1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};

DEVI uint32_t tbl_max(int t) { return t == T_LL ? 35u : t == T_OF ? 31u : 52u; }
DEVI uint32_t tbl_addBits(int t, uint32_t code) { return t == T_LL ? c_LL_bits[code] : t == T_OF ? code : c_ML_bits[code]; }
DEVI uint32_t tbl_maxLog(int t) { return t == T_OF ? 8u : 9u; }

// ZSTD_buildFSETable_body (zstd_decompress_block.pas:436-583) into packed cells.  `norm` holds maxSV+1 counts.
template <typename CellPtr, typename NormPtr, typename NextPtr>
DEVI void dev_build_seq_table(int t, CellPtr cells, NormPtr norm, NextPtr symNext, uint32_t maxSV, uint32_t tableLog)
{
    uint32_t const tableSize = 1u << tableLog;
    uint32_t highThreshold = tableSize - 1;
    for (uint32_t s = 0; s <= maxSV; s++) {
        if (norm[s] == -1) { cells[highThreshold--] = s; symNext[s] = 1; }
        else symNext[s] = (uint16_t)norm[s];
    }
    uint32_t const mask = tableSize - 1, step = (tableSize >> 1) + (tableSize >> 3) + 3;
    uint32_t pos = 0;
    for (uint32_t s = 0; s <= maxSV; s++) {
        int const n = norm[s];
        for (int i = 0; i < n; i++) {
            cells[pos] = s;
            pos = (pos + step) & mask;
            while (pos > highThreshold) pos = (pos + step) & mask;
        }
    }
    for (uint32_t u = 0; u < tableSize; u++) {
        uint32_t const sym = cells[u];
        uint32_t const ns = symNext[sym];
        symNext[sym] = (uint16_t)(ns + 1);
        uint32_t const nb = tableLog - hb32(ns);
        cells[u] = seq_cell(t, (ns << nb) - tableSize, nb, tbl_addBits(t, sym), sym);
    }
}

// Build table `t` from a description of type `type` at `ptr` (`avail` bytes left in the block):
// ZSTD_buildSeqTable (zstd_decompress_block.pas:626-688) without the repeat arm.
// Returns bytes consumed or kFail; sets log.
template <typename CellPtr, typename NormPtr, typename NextPtr>
DEVI uint32_t dev_build_from_desc(int t, uint32_t type, const uint8_t* ptr, uint32_t avail, CellPtr cells, NormPtr norm, NextPtr symNext, uint32_t& log)
{
    if (type == SET_REPEAT) return kFail;   // callers resolve repeat mode to the defining block first
    if (type == SET_RLE) {
        if (avail < 1) return kFail;
        uint32_t const sym = ptr[0];
        if (sym > tbl_max(t)) return kFail;
        cells[0] = seq_cell(t, 0, 0, tbl_addBits(t, sym), sym);
        log = 0;
        return 1;
    }
    if (type == SET_BASIC) {   // predefined distributions (zstd_internal.pas:64-99), tables of zstd_decompress_block.pas:29-125
        uint32_t maxSV;
        if (t == T_LL) { maxSV = 35; log = 6; for (uint32_t s = 0; s <= maxSV; s++) norm[s] = c_LL_defaultNorm[s]; }
        else if (t == T_OF) { maxSV = 28; log = 5; for (uint32_t s = 0; s <= maxSV; s++) norm[s] = c_OF_defaultNorm[s]; }
        else { maxSV = 52; log = 6; for (uint32_t s = 0; s <= maxSV; s++) norm[s] = c_ML_defaultNorm[s]; }
        dev_build_seq_table(t, cells, norm, symNext, maxSV, log);
        return 0;
    }
    // SET_COMPRESSED
    uint32_t maxSV = tbl_max(t), tableLog = 0;
    uint32_t const h = dev_readNCount(norm, maxSV, tableLog, ptr, avail);
    if (h == kFail) return kFail;
    if (tableLog > tbl_maxLog(t)) return kFail;
    dev_build_seq_table(t, cells, norm, symNext, maxSV, tableLog);
    log = tableLog;
    return h;
}

// Size in bytes of the description of table `t` of block `sb` at payload offset `o` (used to walk to a later description).
template <typename NormPtr>
DEVI uint32_t dev_desc_size(int t, uint32_t type, const uint8_t* payload, uint32_t o, uint32_t cSize, NormPtr norm)
{
    if (type == SET_RLE) return o < cSize ? 1 : kFail;
    if (type != SET_COMPRESSED) return 0;
    if (o > cSize) return kFail;
    uint32_t maxSV = tbl_max(t), tableLog = 0;
    return dev_readNCount(norm, maxSV, tableLog, payload + o, cSize - o);
}

// What the first lane of a group found for one table: how the 16 lanes fill it.
enum : uint32_t { TK_FAIL = 0, TK_NORM = 1, TK_RLE = 2, TK_DICT = 3 };

// Description of type `type` -> norm[] (TK_NORM; the table is then built by the group) or a single cell (TK_RLE).
// ZSTD_buildSeqTable (zstd_decompress_block.pas:626-688) without the repeat arm.  Returns bytes consumed or kFail.
template <typename CellPtr, typename NormPtr, typename SrcT>
DEVI uint32_t dev_parse_desc(int t, uint32_t type, SrcT src, uint32_t avail, CellPtr cells, NormPtr norm,
                             uint32_t& kind, uint32_t& maxSV, uint32_t& log, const uint8_t* llBits, const uint8_t* mlBits)
{
    kind = TK_FAIL;
    if (type == SET_REPEAT) return kFail;   // callers resolve repeat mode to the defining block first
    if (type == SET_RLE) {
        if (avail < 1) return kFail;
        uint32_t const sym = src[0];
        if (sym > tbl_max(t)) return kFail;
        cells[0] = seq_cell(t, 0, 0, t == T_LL ? llBits[sym] : t == T_OF ? sym : mlBits[sym], sym);
        log = 0; kind = TK_RLE;
        return 1;
    }
    if (type == SET_BASIC) {   // predefined distributions (zstd_internal.pas:64-99)
        if (t == T_LL) { maxSV = 35; log = 6; for (uint32_t s = 0; s <= maxSV; s++) norm[s] = c_LL_defaultNorm[s]; }
        else if (t == T_OF) { maxSV = 28; log = 5; for (uint32_t s = 0; s <= maxSV; s++) norm[s] = c_OF_defaultNorm[s]; }
        else { maxSV = 52; log = 6; for (uint32_t s = 0; s <= maxSV; s++) norm[s] = c_ML_defaultNorm[s]; }
        kind = TK_NORM;
        return 0;
    }
    maxSV = tbl_max(t);
    uint32_t tableLog = 0;
    uint32_t const h = dev_readNCount(norm, maxSV, tableLog, src, avail);
    if (h == kFail) return kFail;
    if (tableLog > tbl_maxLog(t)) return kFail;
    log = tableLog; kind = TK_NORM;
    return h;
}

// ZSTD_buildFSETable_body (zstd_decompress_block.pas:436-583) by a group of 16 lanes (a DPP row).  The serial
// algorithm spreads the symbols over the table in steps of `step`, skipping the cells above highThreshold, and
// then numbers the cells of each symbol in table order.  Here:
//   1. the first lane lays out the symbols: low-probability ones at the top, a running start for the others;
//   2. lanes fill occ[r] = symbol of the r-th spread step (by symbol), then each takes a 16th of the step
//      sequence, counts its unskipped steps, and after a prefix sum over the lanes writes its symbols;
//   3. each lane counts the symbols in its 16th of the table (rank of every cell inside the chunk), the counts
//      are prefix-summed per symbol over the lanes, and every cell gets nextState = norm + cells of its symbol
//      before it.
// scratch: cum u16[64] | occ u8[512] | cnt u16[16][56]  (2432 bytes; symbols go up to 52)
constexpr uint32_t kCntStride = 56;
constexpr uint32_t kGrpBuildScratch = 128 + 512 + 16 * 2 * kCntStride;
DEVI void grp_build_seq_table(int t, uint32_t* cells, const int16_t* norm, uint32_t maxSV, uint32_t tableLog, uint8_t* scratch, uint32_t gl,
                              const uint8_t* llBits, const uint8_t* mlBits)
{
    // LDS reads are issued in batches of 8 or 16 ahead of the stores that depend on them: the loops below are chains of
    // LDS round trips otherwise (the compiler keeps a load behind every earlier store that might alias it).
    uint16_t* const cum = (uint16_t*)scratch;
    uint8_t* const occ = scratch + 128;
    uint16_t* const cnt = (uint16_t*)(scratch + 128 + 512);
    uint32_t const tableSize = 1u << tableLog, mask = tableSize - 1, step = (tableSize >> 1) + (tableSize >> 3) + 3;
    if (gl == 0) {
        uint32_t high = tableSize - 1, acc = 0;
        for (uint32_t s0 = 0; s0 <= maxSV; s0 += 8) {      // (norm[] has 64 entries: reading past maxSV is harmless)
            int n[8];
#pragma unroll
            for (int j = 0; j < 8; j++) n[j] = norm[s0 + j];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                uint32_t const sym = s0 + j;
                if (sym <= maxSV) {
                    if (n[j] == -1) cells[high--] = sym;
                    else { cum[sym] = (uint16_t)acc; acc += (uint32_t)n[j]; }
                }
            }
        }
        cum[63] = (uint16_t)high;      // (symbols go up to 52)
    }
    lds_sync();
    uint32_t const high = cum[63];
    for (uint32_t s = gl; s <= maxSV; s += 16) {
        int const n = norm[s];
        uint32_t const c0 = cum[s];
        if (n > 0) fill_bytes(occ + c0, 0x01010101u * s, (uint32_t)n);
    }
    lds_sync();
    uint32_t const C = tableSize >> 4, i0 = gl * C;       // tableSize >= 32: C = 2, 4, 8, 16 or 32
    {
        uint32_t local = 0;
        for (uint32_t i = i0; i < i0 + C; i++) local += (((i * step) & mask) <= high) ? 1u : 0u;
        uint32_t incl = local;
        incl += dpp_or0<0x111, 0xF>(incl); incl += dpp_or0<0x112, 0xF>(incl); incl += dpp_or0<0x114, 0xF>(incl); incl += dpp_or0<0x118, 0xF>(incl);
        uint32_t r = incl - local;
        for (uint32_t g = 0; g < C; g += 8) {
            // the next 8 symbols of the spread sequence (at most 8 are used by this group of steps)
            uint32_t const w0 = occ[r] | (occ[r + 1] << 8) | (occ[r + 2] << 16) | ((uint32_t)occ[r + 3] << 24);
            uint32_t const w1 = occ[r + 4] | (occ[r + 5] << 8) | (occ[r + 6] << 16) | ((uint32_t)occ[r + 7] << 24);
            uint64_t w = ((uint64_t)w1 << 32) | w0;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                uint32_t const i = i0 + g + j;
                uint32_t const pos = (i * step) & mask;
                if (g + j < C && pos <= high) { cells[pos] = (uint32_t)w & 255u; w >>= 8; r++; }
            }
        }
    }
    for (uint32_t q = 0; q < kCntStride / 8; q++) *(uint4*)(cnt + kCntStride * gl + 8 * q) = make_uint4(0, 0, 0, 0);
    lds_sync();
    for (uint32_t g = 0; g < C; g += 8) {
        uint32_t sym[8], c[8];
#pragma unroll
        for (int j = 0; j < 8; j++) sym[j] = cells[i0 + (g + j < C ? g + j : 0u)];
#pragma unroll
        for (int j = 0; j < 8; j++) c[j] = cnt[kCntStride * gl + sym[j]];
        // occurrences of the same symbol earlier in this batch
#pragma unroll
        for (int j = 1; j < 8; j++) {
#pragma unroll
            for (int i = 0; i < j; i++) c[j] += (sym[i] == sym[j]) ? 1u : 0u;
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (g + j < C) {
                cnt[kCntStride * gl + sym[j]] = (uint16_t)(c[j] + 1);      // (in order: the last occurrence leaves the final count)
                cells[i0 + g + j] = sym[j] | (c[j] << 8);
            }
        }
    }
    lds_sync();
    for (uint32_t s = gl; s <= maxSV; s += 16) {
        int const n = norm[s];
        uint32_t run = n == -1 ? 1u : (uint32_t)n;
        uint32_t c[16];
#pragma unroll
        for (int l = 0; l < 16; l++) c[l] = cnt[kCntStride * l + s];
#pragma unroll
        for (int l = 0; l < 16; l++) { cnt[kCntStride * l + s] = (uint16_t)run; run += c[l]; }
    }
    lds_sync();
    for (uint32_t g = 0; g < C; g += 8) {
        uint32_t v[8], k[8], ab[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = cells[i0 + (g + j < C ? g + j : 0u)];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint32_t const sym = v[j] & 255u;
            k[j] = cnt[kCntStride * gl + sym];
            ab[j] = t == T_LL ? llBits[sym] : t == T_OF ? sym : mlBits[sym];
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (g + j < C) {
                uint32_t const sym = v[j] & 255u;
                uint32_t const ns = k[j] + (v[j] >> 8);
                uint32_t const nb = tableLog - hb32(ns);
                cells[i0 + g + j] = seq_cell(t, (ns << nb) - tableSize, nb, ab[j], sym);
            }
        }
    }
    lds_sync();
}

// Stage hook: the three predefined tables (ZSTD_buildSeqTable's set_basic arm; the reference holds them as constants,
// zstd_decompress_block.pas:29-125) exactly as the sequence kernel builds them -- same description parser, same
// cooperative builder -- unpacked to {nextState, nbAdditionalBits, nbBits, baseValue}: out[t][cell][4], t = LL, OF, ML.
__global__ __launch_bounds__(64) void k_basic_tables(uint32_t* out)
{
    __shared__ __attribute__((aligned(16))) uint32_t cells[64];
    __shared__ __attribute__((aligned(16))) int16_t norm[64];
    __shared__ __attribute__((aligned(16))) uint8_t bscr[kGrpBuildScratch];
    __shared__ uint8_t sLLbits[36], sMLbits[53];
    __shared__ uint32_t note[3];
    uint32_t const lane = threadIdx.x;
    if (lane < 36) sLLbits[lane] = c_LL_bits[lane];
    if (lane < 53) sMLbits[lane] = c_ML_bits[lane];
    wave_sync();
    if (lane >= 16) return;
    for (int t = 0; t < 3; t++) {
        if (lane == 0) {
            uint32_t kind = TK_FAIL, maxSV = 0, lg = 0;
            dev_parse_desc(t, SET_BASIC, (const uint8_t*)nullptr, 0u, cells, norm, kind, maxSV, lg, sLLbits, sMLbits);
            note[0] = kind; note[1] = maxSV; note[2] = lg;
        }
        lds_sync();
        uint32_t const maxSV = note[1], lg = note[2];
        grp_build_seq_table(t, cells, norm, maxSV, lg, bscr, lane, sLLbits, sMLbits);
        for (uint32_t u = lane; u < 64; u += 16) {
            uint32_t const c = cells[u];
            uint32_t const add = cell_addBits(t, c);
            uint32_t const base = t == T_LL ? c_LL_base[cell_llCode(c)] : t == T_ML ? c_ML_base[cell_mlCode(c)] : (add < 2 ? add : (1u << add) - 3u);
            uint32_t* const o = out + 4 * (64 * t + u);
            bool const live = u < (1u << lg);
            o[0] = live ? (c & 0x1FFu) : 0u; o[1] = live ? add : 0u; o[2] = live ? cell_nbBits(c) : 0u; o[3] = live ? base : 0u;
        }
        lds_sync();
    }
}

// ---------------------------------------------------------------- stage 1: sequences
// value of the lane D places lower in the same row of 16 lanes (own value where there is none)
template <int D>
DEVI uint32_t row_up(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x110 + D, 0xF, 0xF, false); }
// quad permute: lane i of every quad takes the value of lane (QP >> 2 i) & 3 of the quad
template <int QP>
DEVI uint32_t quad_perm(uint32_t x) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, QP, 0xF, 0xF, true); }
// value of lane L of the own quad (4 consecutive lanes)
template <int L>
DEVI uint32_t quad_lane(uint32_t x) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, L * 0x55, 0xF, 0xF, true); }
// lane 15 - i of the row (row_mirror): the first lane of a row receives the last lane's value
DEVI uint32_t row_last_to_first(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x140, 0xF, 0xF, false); }
constexpr int kSeqJobs = 16;                // jobs (blocks) per workgroup
#ifndef ZG_CHAIN_WAVES
#define ZG_CHAIN_WAVES 1
#endif
constexpr int kChainWaves = ZG_CHAIN_WAVES;  // 1: one wave runs the chains of all 16 jobs; 4: four waves, four jobs (16 lanes) each
static_assert(kChainWaves == 1 || kChainWaves == 4, "chain waves");
constexpr int kSeqThreads = 64 * (kChainWaves + 4);   // chain waves: the serial chains, four lanes per job; then four helper waves, 16 lanes per job
constexpr uint32_t kSeqRound = 64;          // sequences per round
constexpr uint32_t kWinUnits = 128;         // circular window of the bitstream, 16-byte units (+ 1 mirror unit)
constexpr uint32_t kWinAhead = 80;          // units kept loaded below the cursor of the round before last
constexpr uint32_t kSeqJobWords = 512 + 512 + 256 + 4 * (kWinUnits + 1) + 2 * 2 * kSeqRound;   // LL | ML | OF cells, window, two rings of chain notes
// A chain note is 8 bytes per sequence, 16 bits from each of a block's four chain lanes: the LL, ML and OF states the
// sequence was decoded from (the helpers read the cells again) and the low 16 bits of the cursor before it (a round moves
// the cursor by 64 x 89 bits at most, so the helpers can complete it from the cursor at the start of the round).
constexpr uint32_t kRepIdxShift = 29;

DEVI uint32_t rep_resolve(uint32_t x, const uint32_t* repIn)
{
    if (!(x & kSymbolicRep)) return x;
    uint32_t const v = repIn[(x >> kRepIdxShift) & 3];
    uint32_t const d = x & ((1u << kRepIdxShift) - 1);
    return v > d ? v - d : 1;
}

// A repeat-offset stack entry is either a concrete offset or "slot k of some earlier stack, minus d
// (never below 1)".  Substituting the earlier stack (p0,p1,p2) composes two such descriptions; the
// operation is associative, which lets the stack after every sequence be found with a prefix scan.
DEVI uint32_t rep_subst(uint32_t e, uint32_t p0, uint32_t p1, uint32_t p2)
{
    uint32_t const k = (e >> kRepIdxShift) & 3, d = e & ((1u << kRepIdxShift) - 1);
    uint32_t const x = k == 0 ? p0 : k == 1 ? p1 : p2;
    uint32_t const r = (x & kSymbolicRep) ? x + d : (x > d ? x - d : 1u);
    return (e & kSymbolicRep) ? r : e;
}
struct Rep3 { uint32_t a, b, c; };
DEVI Rep3 rep_compose(Rep3 first, Rep3 then)   // apply `first`, then `then`
{
    Rep3 r;
    r.a = rep_subst(then.a, first.a, first.b, first.c);
    r.b = rep_subst(then.b, first.a, first.b, first.c);
    r.c = rep_subst(then.c, first.a, first.b, first.c);
    return r;
}
constexpr uint32_t kSlot0 = kSymbolicRep | (0u << kRepIdxShift), kSlot1 = kSymbolicRep | (1u << kRepIdxShift), kSlot2 = kSymbolicRep | (2u << kRepIdxShift);

// Sequence records written by this stage, 16 bytes each, nbSeq + 1 per block:
//   {offset (possibly symbolic), output position, literal position, litLength}, positions block-relative;
//   the extra last record carries the end positions, so matchLength = next.outPos - outPos - litLength.
//
// A workgroup of five waves decodes 16 blocks in rounds of 64 sequences:
//   wave 0, lane j    the serial chain of block j only: 3 table cells -> bit widths -> bit cursor -> 3 next states
//                     (ZSTD_decodeSequence's state half, zstd_decompress_block.pas:1092-1101, :1220-1225); the cells
//                     and the cursor of every sequence go into a ring in LDS
//   waves 1-4         16 lanes per block, 4 sequences per lane, one round behind the chain: extra bits, base
//                     values, repeat-offset resolution by prefix scan, positions by prefix sum, coalesced record
//                     stores (:1105-1180); they also build the tables and keep the bitstream window in LDS filled
// The bitstream sits in a circular LDS window of 2 KB per block, addressed by absolute bit position Q counted from
// a 16-byte aligned origin 2 KB below the stream's first byte; unit u (16 bytes) lives in slot u mod 128, and slot 0
// is mirrored behind slot 127 so that a peek never wraps.  While round k+1 is decoded, the helpers that work on
// round k fetch the units below (cursor after round k) - 80 units; the chain stops a round early in the
// (pathological) case that it would pass the loaded floor.
// Whether zstd 1.4.8 decodes a block's sequences with its prefetching loop (ZSTD_decompressSequencesLong) rather than
// the regular one: only in frames with a window above 2^24, for blocks of more than 4 sequences whose offset table has
// at least 7/256 of its cells with more than 22 additional bits (ZSTD_getLongOffsetsShare, zstd_decompress_block.pas:1460-1488,
// dispatch :1527-1541).  The two loops regenerate the same bytes from a sound stream and differ on a damaged one: the
// prefetching loop stops with corruption_detected once the stream has run dry with sequences left (:1376-1400), and does
// not check that the stream was used up.
template <typename CellPtr>
DEVI bool prefetching_loop_148(uint32_t longWindow, uint32_t nbSeq, CellPtr tOF, uint32_t logOF)
{
    if (!longWindow || nbSeq <= 4) return false;
    uint32_t total = 0;
    for (uint32_t u = 0; u < (1u << logOF); u++) total += cell_addBits(T_OF, tOF[u]) > 22 ? 1u : 0u;
    return (total << (8 - logOF)) >= 7;
}

struct SeqCtl {            // per block, in LDS
    uint32_t fail, nbSeq, logs, Qinit, Qstart;      // logs = logLL | logOF << 8 | logML << 16
};

DEVI void wg_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// 64 bits below bit Q of a circular window (tHi = bits [Q-32, Q))
DEVI void peek64c(const uint32_t* win, uint32_t Q, uint32_t& tHi, uint32_t& tLo)
{
    uint32_t const d = ((Q >> 5) - 2) & (4 * kWinUnits - 1);
    uint32_t const a0 = win[d], a1 = win[d + 1], a2 = win[d + 2];
    tHi = __builtin_amdgcn_alignbit(a2, a1, Q);
    tLo = __builtin_amdgcn_alignbit(a1, a0, Q);
}

// 16-byte unit at addr, kept inside [floor, ceil] (whatever lies outside is never interpreted)
DEVI uint4 ld_unit(uintptr_t addr, uintptr_t floor, uintptr_t ceil) { return ld_unit(addr <= ceil ? addr : ceil, floor); }

// kDictTables: every job of the launch takes all three tables from the dictionary (small dictionary-compressed records:
// the tables are then copied into LDS once per workgroup instead of once per block, which leaves room for two workgroups
// -- two chain waves -- on a compute unit).
// It is launched with eight waves, of which the last three end at once: a compute unit was seen to hold only ONE workgroup
// of five waves with this much LDS (72 KB; 48 KB: two, not three), but two of eight or of four waves (tools/census).
constexpr int kSeqLaunchThreadsDict = 64 * 8;
constexpr uint32_t kSeqTabWords = 512 + 512 + 256;
constexpr uint32_t kSeqOwnWords = kSeqJobWords - kSeqTabWords;      // window and rings of a job
template <bool kDictTables>
__global__ __launch_bounds__(kDictTables ? kSeqLaunchThreadsDict : kSeqThreads) void k_sequences(BatchParams p)
{
    if (kDictTables && threadIdx.x >= (uint32_t)kSeqThreads) return;      // (waves that only make up the launch shape; they never reach a barrier)
    __shared__ __attribute__((aligned(16))) uint32_t lds[kDictTables ? kSeqTabWords + kSeqJobs * kSeqOwnWords : kSeqJobs * kSeqJobWords];
    __shared__ uint32_t sLLbase[36], sMLbase[53];
    __shared__ uint8_t sLLbits[36], sMLbits[53];
    __shared__ SeqCtl ctl[kSeqJobs];
    __shared__ uint2 pub[2][kSeqJobs];         // per round parity and block: {cursor after the round, sequences decoded in it}
    __shared__ __attribute__((aligned(16))) uint32_t activeFlag[2][4];      // per round parity and chain wave: some block of its still has sequences to decode in that round
    __shared__ uint32_t zeroCell;              // the "table" of a block's fourth chain lane: its cell adds nothing to the sum of the quad
    uint32_t const tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
#ifdef ZG_STAMPS
    uint64_t const k0 = __builtin_readcyclecounter(); uint64_t kSetup = 0, kLoop = 0, kRounds = 0;
#endif
    if (tid < 36) { sLLbase[tid] = c_LL_base[tid]; sLLbits[tid] = c_LL_bits[tid]; }
    if (tid < 53) { sMLbase[tid] = c_ML_base[tid]; sMLbits[tid] = c_ML_bits[tid]; }
    if (tid < 8) activeFlag[tid >> 2][tid & 3] = (tid < (uint32_t)kChainWaves) ? 1u : 0u;
    if (tid == 0) zeroCell = 0;
    wg_barrier();

    // ---------------- helper side set-up (waves 1-4): tables, window, per-block constants
    uint32_t const gl = lane & 15, lead = lane & ~15u;
    bool const chain = wave < (uint32_t)kChainWaves, helper = !chain;
    // (with four chain waves only the first 16 lanes of each run chains: fewer addresses per LDS read, fewer bank conflicts)
    bool const chainLane = chain && (kChainWaves == 1 || lane < 16);
    uint32_t const slot = helper ? (wave - kChainWaves) * 4 + (lane >> 4)
                                 : (kChainWaves == 1 ? (lane >> 2) : wave * 4 + ((lane >> 2) & 3));       // block slot inside the workgroup
    uint32_t const role = lane & 3;            // chain wave: which of a block's four chain lanes this is (LL, ML, OF, cursor)
    uint32_t const job = blockIdx.x * kSeqJobs + slot;
    bool const hasJob = job < n_seq_jobs(p);
    uint32_t* const tLL = kDictTables ? lds : lds + slot * kSeqJobWords;
    uint32_t* const tML = tLL + 512;
    uint32_t* const tOF = tML + 512;
    uint32_t* const win = kDictTables ? lds + kSeqTabWords + slot * kSeqOwnWords : tOF + 256;
    if (kDictTables) {
        // the dictionary's three tables, once for the workgroup (16 bytes per thread and pass; whole arrays: cells past a
        // table's size are never addressed)
        const DevDict* const dd = p.dict;
        for (uint32_t i = tid; i < kSeqTabWords / 4; i += kSeqThreads) {
            const uint4* const from = i < 128 ? (const uint4*)dd->ll + i : i < 256 ? (const uint4*)dd->ml + (i - 128) : (const uint4*)dd->of + (i - 256);
            ((uint4*)lds)[i] = *from;
        }
    }
    uint32_t* const ring0 = win + 4 * (kWinUnits + 1);
    uint32_t const b = hasJob ? p.seqJobs[job] : 0;

    // helper-side per-block state
    DevBlock blk = {};
    uint32_t failU = 1;
    uint32_t Qstart = 0, Qend = 0;             // bit position of the stream's first bit; cursor after the last round seen
    uintptr_t origin = 0, floorAddr = (uintptr_t)p.src, ceilAddr = (uintptr_t)p.src;
    uint32_t loadedFrom = 0;                   // lowest unit already in the window
    if (helper && hasJob) {
        blk = p.blocks[b];
        const uint8_t* const payload = p.src + blk.srcOff;
        uint32_t off = blk.seqHdrOff;
        uint32_t logsPacked = 0;
        {
            // scratch while the tables are built: where the window and the rings will be
            uint8_t* const scr = (uint8_t*)win;
            uint8_t* const hdr = scr;                              // 256 staged bytes of the table descriptions
            int16_t* const norm = (int16_t*)(scr + 256);           // 64 counts
            uint32_t* const note = (uint32_t*)(scr + 384);         // {kind, maxSV, log, consumed} from the first lane
            uint8_t* const bscr = scr + 512;                       // grp_build_seq_table scratch
            static_assert(512 + kGrpBuildScratch <= 4 * (4 * (kWinUnits + 1) + 2 * 2 * kSeqRound), "table scratch fits");
            uint32_t const hdrAvail = off < blk.cSize ? blk.cSize - off : 0;
            uint32_t const nStaged = hdrAvail < 256 ? hdrAvail : 256;
            if (!kDictTables) {
                grp_stage_bytes(hdr, payload + off, nStaged, gl);
                lds_sync();
            }
#ifdef ZG_STAMPS
            if (gl == 0 && p.dbg) p.dbg[8 * (size_t)p.nBlocks + 8 * (size_t)b + 5] = __builtin_readcyclecounter() - k0;
#endif
            bool fail = false;
            uint32_t done = 0;          // description bytes consumed so far
            if (kDictTables) logsPacked = p.dict->llLog | (p.dict->ofLog << 8) | (p.dict->mlLog << 16);
#pragma unroll 1      // (one copy of this code run three times: it is executed once per block, and instruction fetch is what it costs)
            for (int t = 0; t < (kDictTables ? 0 : 3); t++) {
                uint32_t const type = t == T_LL ? blk.llType : t == T_OF ? blk.ofType : blk.mlType;
                int32_t const srcB = t == T_LL ? blk.llSrc : t == T_OF ? blk.ofSrc : blk.mlSrc;
                uint32_t* const cells = t == T_LL ? tLL : t == T_OF ? tOF : tML;
#ifdef ZG_STAMPS
                uint64_t const kp0 = __builtin_readcyclecounter();
#endif
                if (gl == 0) {
                    uint32_t kind = TK_FAIL, maxSV = 0, lg = 0, used = 0;
                    if (fail) {
                    } else if (type != SET_REPEAT) {
                        if (off + done <= blk.cSize) {
                            HdrBytes hb;
                            hb.lds = hdr + (done < nStaged ? done : nStaged); hb.mem = payload + off + done; hb.nStaged = done < nStaged ? nStaged - done : 0;
                            uint32_t const h = dev_parse_desc(t, type, hb, blk.cSize - off - done, cells, norm, kind, maxSV, lg, sLLbits, sMLbits);
                            if (h != kFail) used = h; else kind = TK_FAIL;
                        }
                    } else if (srcB == SRC_DICT) {
                        kind = TK_DICT;
                        lg = t == T_LL ? p.dict->llLog : t == T_OF ? p.dict->ofLog : p.dict->mlLog;
                    } else if (srcB != SRC_NONE) {
                        // rebuild from the block that defined the table: walk its descriptions up to table t
                        DevBlock const sb = p.blocks[srcB];
                        const uint8_t* const sp = p.src + sb.srcOff;
                        uint32_t o = sb.seqHdrOff;
                        bool bad = false;
                        for (int u = 0; u < t && !bad; u++) {
                            uint32_t const ut = u == T_LL ? sb.llType : sb.ofType;
                            uint32_t const sz = dev_desc_size(u, ut, sp, o, sb.cSize, norm);
                            if (sz == kFail) bad = true; else o += sz;
                        }
                        if (!bad && o <= sb.cSize) {
                            uint32_t const stype = t == T_LL ? sb.llType : t == T_OF ? sb.ofType : sb.mlType;
                            if (dev_parse_desc(t, stype, sp + o, sb.cSize - o, cells, norm, kind, maxSV, lg, sLLbits, sMLbits) == kFail) kind = TK_FAIL;
                        }
                    }
                    note[0] = kind; note[1] = maxSV; note[2] = lg; note[3] = used;
                }
                lds_sync();
#ifdef ZG_STAMPS
                if (gl == 0 && p.dbg) p.dbg[8 * (size_t)p.nBlocks + 8 * (size_t)b + 7] += __builtin_readcyclecounter() - kp0;
#endif
                uint32_t const kind = note[0], maxSV = note[1], lg = note[2];
                done += note[3];
                if (kind == TK_FAIL) fail = true;
                else if (kind == TK_NORM) grp_build_seq_table(t, cells, norm, maxSV, lg, bscr, gl, sLLbits, sMLbits);
                else if (kind == TK_DICT) {
                    // (16 bytes per lane and pass, the loads of eight passes -- a whole table of 512 cells -- ahead of the stores;
                    // source and destination hold at least 256 cells and are 16-byte aligned: a short table is copied rounded up)
                    const uint4* const dc = (const uint4*)(t == T_LL ? p.dict->ll : t == T_OF ? p.dict->of : p.dict->ml);
                    uint4* const to = (uint4*)cells;
                    uint32_t const units = ((4u << lg) + 15) >> 4;
                    // (a lane with no piece left in a pass copies the table's last piece again: the pieces stay in registers)
                    uint32_t const last = units - 1;
                    uint32_t const i0 = min(gl, last), i1 = min(gl + 16, last), i2 = min(gl + 32, last), i3 = min(gl + 48, last);
                    uint4 const q0 = dc[i0], q1 = dc[i1], q2 = dc[i2], q3 = dc[i3];
                    to[i0] = q0; to[i1] = q1; to[i2] = q2; to[i3] = q3;
                    if (units > 64) {
                        uint32_t const i4 = min(gl + 64, last), i5 = min(gl + 80, last), i6 = min(gl + 96, last), i7 = min(gl + 112, last);
                        uint4 const q4 = dc[i4], q5 = dc[i5], q6 = dc[i6], q7 = dc[i7];
                        to[i4] = q4; to[i5] = q5; to[i6] = q6; to[i7] = q7;
                    }
                }
                logsPacked |= lg << (t == T_LL ? 0 : t == T_OF ? 8 : 16);
                lds_sync();      // the note and the scratch are free again
            }
#ifdef ZG_STAMPS
            if (gl == 0 && p.dbg) p.dbg[8 * (size_t)p.nBlocks + 8 * (size_t)b + 6] = __builtin_readcyclecounter() - k0;
#endif
            off += done;
            if (!fail && off >= blk.cSize) fail = true;                     // BIT_initDStream on an empty stream
            if (!fail && payload[blk.cSize - 1] == 0) fail = true;          // no end mark
            failU = fail ? 1u : 0u;
        }
        if (!failU) {
            const uint8_t* const sStart = payload + off;
            uint32_t const sSize = blk.cSize - off;
            uint32_t const bitsLeft = sSize * 8 - (8 - hb32(sStart[sSize - 1]));
            origin = ((uintptr_t)sStart & ~(uintptr_t)15) - 16 * kWinUnits;
            Qstart = 8 * (uint32_t)((uintptr_t)sStart - origin);
            Qend = Qstart + bitsLeft;
            ceilAddr = ((uintptr_t)sStart + sSize - 1) & ~(uintptr_t)15;
            // first fill: units (Qend >> 7) - 80 .. the unit of the stream's last byte
            loadedFrom = (Qend >> 7) - kWinAhead;
            uint4 v[6];
#pragma unroll
            for (int k = 0; k < 6; k++) v[k] = ld_unit(origin + 16 * (uintptr_t)(loadedFrom + gl + 16 * k), floorAddr, ceilAddr);
#pragma unroll
            for (int k = 0; k < 6; k++) {
                uint32_t const u = loadedFrom + gl + 16 * k;
                if (u <= ((Qend - 1) >> 7)) {
                    *(uint4*)(win + 4 * (u & (kWinUnits - 1))) = v[k];
                    if ((u & (kWinUnits - 1)) == 0) *(uint4*)(win + 4 * kWinUnits) = v[k];
                }
            }
            if (gl == 0) { ctl[slot].fail = 0; ctl[slot].nbSeq = blk.nbSeq; ctl[slot].logs = logsPacked; ctl[slot].Qinit = Qend; ctl[slot].Qstart = Qstart; }
        } else if (gl == 0) {
            ctl[slot].fail = 1; ctl[slot].nbSeq = 0; ctl[slot].logs = 0; ctl[slot].Qinit = 0; ctl[slot].Qstart = 0;
        }
    } else if (helper && gl == 0) {
        ctl[slot].fail = 1; ctl[slot].nbSeq = 0; ctl[slot].logs = 0; ctl[slot].Qinit = 0; ctl[slot].Qstart = 0;
    }
    wg_barrier();

    // ---------------- chain side set-up (wave 0): four lanes per block, one per table plus one that records the cursor
    // (its own "cell" is a zero word).  Every lane keeps one FSE state; what the lanes need of each other's cells comes
    // through DPP quad permutes.
    uint32_t st = 0;                           // this lane's FSE state
    uint32_t Q = 0, e1 = 0, e2 = 0;            // cursor; cursor after the previous round and after the one before
    uint32_t remaining = 0;
    int dryBelow = 0;                          // a cursor below this is past the stream start by more than the last state update may read
    typedef const __attribute__((address_space(3))) uint32_t* lds_u32;
    typedef __attribute__((address_space(3))) uint16_t* lds_u16w;
    uint32_t const tabL = (uint32_t)(uintptr_t)(lds_u32)(role == 0 ? tLL : role == 1 ? tML : role == 2 ? tOF : &zeroCell);    // LDS byte address of this lane's table
    if (chainLane) {
        SeqCtl const c = ctl[slot];
        if (!c.fail) {
            remaining = c.nbSeq;
            dryBelow = (int)c.Qstart - 26;
            Q = e1 = e2 = c.Qinit;
            // ZSTD_initFseState x3, order LL, OF, ML (:1266-1268), <= 26 bits
            uint32_t const logLL = c.logs & 255, logOF = (c.logs >> 8) & 255, logML = c.logs >> 16;
            uint32_t tHi, tLo;
            peek64c(win, Q, tHi, tLo);
            st = role == 0 ? take(tHi, 0, logLL) : role == 1 ? take(tHi, logLL + logOF, logML) : role == 2 ? take(tHi, logLL, logOF) : 0u;
            Q -= logLL + logOF + logML;
        }
    }

    uint32_t mask1 = role >= 1 ? 0xFFFFFFFFu : 0u, mask2 = role >= 2 ? 0xFFFFFFFFu : 0u;
    asm volatile("" : "+v"(mask1), "+v"(mask2));       // plain data to the compiler: an AND with a DPP operand, not a select

    // helper-side running state; carried in the group's first lane only: stack after the sequences done so far
    // (relative to the block's incoming stack) and the positions, saturating at 2^31-1 (frames regenerate < 2 GiB,
    // so a saturated position always fails the capacity check)
    Rep3 rep = {kSlot0, kSlot1, kSlot2};
    uint32_t outPos = 0, litPos = 0;
    uint32_t const nbSeq = blk.nbSeq;
    uint32_t nValid = nbSeq, seqDone = 0;
    bool dead = failU != 0;                    // nothing (more) to do for this block on the helper side
    uint4* const out = (uint4*)p.seqBuf + blk.seqBufOff;
    uint4* const trash = out + nbSeq + 1;      // 64 spare records per block take the stores of lanes without a sequence

#ifdef ZG_STAMPS
    kSetup = __builtin_readcyclecounter() - k0;
#endif
    bool chainAct = true;       // the chain wave's own copy of the flag it publishes (no LDS round trip at the top of its rounds)
    for (uint32_t k = 0;; k++) {
        uint32_t const par = k & 1;
        bool act;
        if (kChainWaves == 1) { if (chain) act = chainAct; else act = activeFlag[par][0] != 0; }
        else { uint4 const f = *(const uint4*)activeFlag[par]; act = (f.x | f.y | f.z | f.w) != 0; }
        if (chain) {
            // ---- the serial chains, round k
            if (act && chainLane) {
                uint32_t n = 0;
                uint32_t* const ring = ring0 + par * 2 * kSeqRound;
                if (remaining) {
                    // lowest bit the window is known to hold, plus what 8 sequences and a 96-bit peek can reach (as a bound on
                    // T = cursor - 95, the variable the loop carries)
                    uint32_t const Tlim = ((e2 >> 7) - kWinAhead) * 128 + 96 + 8 * 89 - 95;
                    uint32_t i = 0;
                    uint32_t winL = (uint32_t)(uintptr_t)(lds_u32)win;     // LDS byte addresses, opaque to the compiler: keeps the
                    uint32_t ringL = (uint32_t)(uintptr_t)(lds_u32)ring + 2 * role;    // accesses at small immediate offsets of one address
                    asm volatile("" : "+v"(winL), "+v"(ringL));
                    // Software pipeline: the cell of the next sequence is requested as soon as its state is known, and the
                    // bookkeeping (ring entry, cursor, window request for the next sequence) runs while that read is in flight.
                    auto win_read = [&](uint32_t T, uint32_t& a0, uint32_t& a1, uint32_t& a2, uint32_t& a3) {
                        uint32_t word = __builtin_amdgcn_ubfe(T, 5, 9);
                        asm volatile("" : "+v"(word));      // (a field extract and a shift-add: kept from being turned into three instructions)
                        lds_u32 const w = (lds_u32)(uintptr_t)(winL + 4 * word);
                        a0 = w[0]; a1 = w[1]; a2 = w[2]; a3 = w[3];
                    };
                    // 96 bits below the cursor (top at Q + 1, so that the state bits of any sequence start inside them)
                    uint32_t T = Q - 95, a0, a1, a2, a3;
                    uint32_t const Tround = T;
                    win_read(T, a0, a1, a2, a3);
                    uint32_t c = *(lds_u32)(uintptr_t)(tabL + 4 * st);
#ifdef ZG_STAMPS
                    uint64_t const kl0 = __builtin_readcyclecounter();
#endif
                    uint32_t ringP = ringL;        // ring entry of the group's first sequence: the eight stores of a group go to it + constants
                    for (; i < kSeqRound; i += 8) {
                        if (T < Tlim) break;
                        // Past the block's last sequence the chain runs on over bits that mean nothing, inside the window
                        // and inside the tables (a state never leaves its table); those ring entries are ignored, and
                        // entry n records the cursor (less 95) after the last real sequence.
#pragma unroll
                        for (uint32_t j = 0; j < 8; j++) {
                            // sum of the block's three cells: bit counts of the sequence
                            // (the fourth lane's cell is zero and its state stays zero)
                            uint32_t S = c + quad_perm<0xB1>(c);        // lanes 1 0 3 2
                            asm volatile("" : "+v"(S));
                            S += quad_perm<0x4E>(S);                    // lanes 2 3 0 1
                            // state bits, order LL, ML, OF (:1220-1225): this lane's lie under those of the tables before it
                            uint32_t const stPrev = st;
                            uint32_t const nb = cell_nbBits(c);
                            uint32_t const ng = __builtin_amdgcn_ubfe(c, 14, 5);
                            uint32_t const o = ng + (quad_perm<0x90>(ng) & mask1) + (quad_perm<0x40>(ng) & mask2);   // + the lane below, + two lanes below; mod 32
                            // the 32 bits under the additional bits: bit R = 63 - (their number) of (t2:t1:t0) upwards
                            uint32_t const R = S >> 26;
                            bool const up = (int32_t)S < 0;         // R >= 32
                            __builtin_amdgcn_sched_barrier(0);      // (what needs the cell only comes first: the window words were requested after it)
                            uint32_t const t2 = __builtin_amdgcn_alignbit(a3, a2, T), t1 = __builtin_amdgcn_alignbit(a2, a1, T), t0 = __builtin_amdgcn_alignbit(a1, a0, T);
                            uint32_t const h3 = __builtin_amdgcn_alignbit(up ? t2 : t1, up ? t1 : t0, R);
                            // a cell's nextState has its low nbBits clear
                            st = (c & 0x1FF) | __builtin_amdgcn_ubfe(h3, o, nb);
                            uint32_t const cn = *(lds_u32)(uintptr_t)(tabL + 4 * st);
                            __builtin_amdgcn_sched_barrier(0);
                            uint32_t const Tprev = T;
                            uint32_t Tm = T + R + 0xFFFFFFC1u;      // additional bits: 63 - R
                            asm volatile("" : "+v"(Tm));            // (one three-operand add)
                            T = Tm - __builtin_amdgcn_ubfe(S, 21, 5);
                            win_read(T, a0, a1, a2, a3);
                            *(lds_u16w)(uintptr_t)(ringP + 8 * j) = (uint16_t)(role == 3 ? Tprev : stPrev);
                            c = cn;
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        ringP += 8 * 8;
                        asm volatile("" : "+v"(ringP));
                    }
#ifdef ZG_STAMPS
                    kLoop += __builtin_readcyclecounter() - kl0; kRounds++;
#endif
                    n = i < remaining ? i : remaining;
                    Q = (n < i ? Tround - ((Tround - (ring[2 * n + 1] >> 16)) & 0xFFFFu) : T) + 95;
                    remaining -= n;
                    // the stream ran dry: some sequence of this round is flagged by the helpers, nothing after it counts
                    if ((int)Q < dryBelow) remaining = 0;
                    e2 = e1; e1 = Q;
                }
                if (role == 0) pub[par][slot] = make_uint2(Q, n);
                bool const anyLeft = __any(remaining != 0);
                if (lane == 0) activeFlag[par ^ 1][wave] = anyLeft ? 1u : 0u;
                chainAct = anyLeft;
            }
        } else if (helper && k >= 1 && !dead) {
            // ---- helpers, round k-1
            uint32_t const hp = par ^ 1;
            uint2 const pb = pub[hp][slot];
            uint32_t const n = pb.y;
            const uint32_t* const ring = ring0 + hp * 2 * kSeqRound;
            if (n) {
                uint32_t const base = seqDone;
                bool const more = base + n < nbSeq;
                uint32_t const Tref = Qend - 95;      // at or above every cursor note of the round, by less than 2^16
                // units the chain may need two rounds from now: down to (cursor after this round) - 80 units
                Qend = pb.x;
                uint32_t const newFrom = (Qend >> 7) - kWinAhead;     // <= loadedFrom; at most 46 units below it
                uint4 v[3];
#pragma unroll
                for (int q = 0; q < 3; q++) v[q] = ld_unit(origin + 16 * (uintptr_t)(newFrom + gl + 16 * q), floorAddr, ceilAddr);
                // ---- sequences 4*gl .. 4*gl+3 of this round
                uint32_t ll[4], len[4], topA[4];
                Rep3 acc = {kSlot0, kSlot1, kSlot2};    // stack transformation of this lane's sequences so far
                uint32_t firstBad = kSeqRound;          // first sequence of this lane whose extra bits ran past the stream start
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    uint32_t const i = 4 * gl + u;
                    uint2 const r = *(const uint2*)(ring + 2 * i);
                    // (masked: the notes of lanes past the round's last sequence may be anything)
                    uint32_t const cLL = tLL[r.x & 0x1FFu], cML = tML[__builtin_amdgcn_ubfe(r.x, 16, 9)], cOF = tOF[r.y & 0xFFu];
                    uint32_t const Qs = Tref - ((Tref - (r.y >> 16)) & 0xFFFFu) + 95;
                    bool const valid = i < n;
                    uint32_t tHi, tLo;
                    peek64c(win, Qs, tHi, tLo);
                    uint32_t const n1 = cell_addBits(T_OF, cOF), n2 = cell_addBits(T_ML, cML), n3 = cell_addBits(T_LL, cLL);
                    uint32_t const ofRaw = take(tHi, 0, n1);
                    uint32_t const h1 = shl_hi(tHi, tLo, n1);
                    uint32_t const llCode = cell_llCode(cLL), mlCode = cell_mlCode(cML);
                    uint32_t const lli = sLLbase[llCode < 36 ? llCode : 0] + take(h1, n2, n3);
                    uint32_t const mli = sMLbase[mlCode < 53 ? mlCode : 0] + take(h1, 0, n2);
                    if (valid && (int)(Qs - (n1 + n2 + n3)) < (int)Qstart && firstBad == kSeqRound) firstBad = i;
                    ll[u] = valid ? lli : 0;
                    len[u] = valid ? lli + mli : 0;
                    // repeat-offset step of this sequence as a stack transformation (:1105-1161):
                    //   real offset v: (v, s0, s1);  repeat idx 0: (s0, s1, s2), 1: (s1, s0, s2), 2: (s2, s0, s1), 3: (s0 - 1, s0, s1)
                    bool const isReal = n1 > 1;
                    uint32_t const real = ((1u << n1) - 3u) + ofRaw;
                    uint32_t const idx = (n1 ? 1u + ofRaw : 0u) + (llCode == 0 ? 1u : 0u);
                    uint32_t const ta = !valid ? kSlot0
                                      : isReal ? (real > 0x7FFFFFFFu ? 0x7FFFFFFFu : real)      // anything that large is beyond every window: kept invalid
                                               : (idx == 0 ? kSlot0 : idx == 1 ? kSlot1 : idx == 2 ? kSlot2 : (kSlot0 | 1u));
                    bool const second0 = valid && (isReal || idx >= 1);     // new second entry: old first (else old second)
                    bool const third1 = valid && (isReal || idx >= 2);      // new third entry: old second (else old third)
                    Rep3 nx;
                    nx.a = rep_subst(ta, acc.a, acc.b, acc.c);
                    nx.b = second0 ? acc.a : acc.b;
                    nx.c = third1 ? acc.b : acc.c;
                    acc = nx;
                    topA[u] = acc.a;        // after every sequence the top of the stack is its offset
                }
                // lane totals; the group's first lane puts the carried stack and positions in front, so the inclusive
                // scan over the 16 lanes yields values relative to the block start
                uint32_t sumLen = len[0] + len[1] + len[2] + len[3], sumLL = ll[0] + ll[1] + ll[2] + ll[3];
                Rep3 incl = acc;
                if (gl == 0) { incl = rep_compose(rep, acc); sumLen += outPos; sumLL += litPos; }
                uint32_t inclLen = sumLen, inclLL = sumLL;
#define ZG_SCAN_STEP(D)                                                                                                  \
                {                                                                                                        \
                    Rep3 o;                                                                                              \
                    o.a = row_up<D>(incl.a); o.b = row_up<D>(incl.b); o.c = row_up<D>(incl.c);                           \
                    uint32_t const oLen = row_up<D>(inclLen), oLL = row_up<D>(inclLL);                                   \
                    if (gl >= D) { incl = rep_compose(o, incl); inclLen += oLen; inclLL += oLL; }                        \
                }
                ZG_SCAN_STEP(1) ZG_SCAN_STEP(2) ZG_SCAN_STEP(4) ZG_SCAN_STEP(8)
#undef ZG_SCAN_STEP
                // state before this lane's first sequence = inclusive value of the lane before it (the carry itself in the first lane)
                Rep3 ex;
                ex.a = row_up<1>(incl.a); ex.b = row_up<1>(incl.b); ex.c = row_up<1>(incl.c);
                uint32_t po = row_up<1>(inclLen), pl = row_up<1>(inclLL);     // < 2^31 + 64 * 2^18: no wrap
                if (gl == 0) { ex = rep; po = outPos; pl = litPos; }
                uint4 rec[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    rec[u] = make_uint4(rep_subst(topA[u], ex.a, ex.b, ex.c),
                                        po > 0x7FFFFFFFu ? 0x7FFFFFFFu : po, pl > 0x7FFFFFFFu ? 0x7FFFFFFFu : pl, ll[u]);
                    po += len[u]; pl += ll[u];
                }
                // records out: 64 bytes per lane, 1 KB per group, coalesced; every lane stores four records (lanes without
                // a sequence into the spare area), so that the number of stores in flight behind the window loads is fixed
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    uint32_t const i = 4 * gl + u;
                    uint4* const dstRec = i < n ? out + base + i : trash + i;
                    *dstRec = rec[u];
                }
                // first bad sequence of the round, if any
                uint32_t const badLanes = (uint32_t)(__ballot(firstBad != kSeqRound) >> lead) & 0xFFFFu;
                if (badLanes) {
                    // the stream ran dry inside this round: positions stop at the first bad sequence
                    uint32_t const src = lead + (uint32_t)__builtin_ctz(badLanes);
                    uint32_t const bad = __shfl(firstBad, src), w = bad & 3;
                    uint32_t const bo = w == 0 ? rec[0].y : w == 1 ? rec[1].y : w == 2 ? rec[2].y : rec[3].y;
                    uint32_t const bl = w == 0 ? rec[0].z : w == 1 ? rec[1].z : w == 2 ? rec[2].z : rec[3].z;
                    outPos = __shfl(bo, src); litPos = __shfl(bl, src);
                    nValid = base + bad;
                    dead = true;
                } else {
                    // carry for the next round: inclusive values of the last lane, handed to the first
                    rep.a = row_last_to_first(incl.a); rep.b = row_last_to_first(incl.b); rep.c = row_last_to_first(incl.c);
                    outPos = row_last_to_first(inclLen); litPos = row_last_to_first(inclLL);
                    outPos = outPos > 0x7FFFFFFFu ? 0x7FFFFFFFu : outPos;
                    litPos = litPos > 0x7FFFFFFFu ? 0x7FFFFFFFu : litPos;
                    seqDone = base + n;
                    if (!more) dead = true;
                }
                // window units for the chain, two rounds ahead (nobody reads these slots now)
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    uint32_t const u = newFrom + gl + 16 * q;
                    if (u < loadedFrom) {
                        *(uint4*)(win + 4 * (u & (kWinUnits - 1))) = v[q];
                        if ((u & (kWinUnits - 1)) == 0) *(uint4*)(win + 4 * kWinUnits) = v[q];
                    }
                }
                loadedFrom = newFrom;
            }
        }
        wg_barrier();
        if (!act || k > (1u << 16)) break;     // (every round with work left decodes at least 8 sequences of each unfinished block)
    }
#ifdef ZG_STAMPS
    if (chain && hasJob && role == 0 && p.dbg) {      // diagnostic build: where the chain wave's cycles went
        uint64_t* const q = p.dbg + 8 * (size_t)p.nBlocks + 8 * (size_t)b;
        q[0] = kSetup; q[1] = kLoop; q[2] = __builtin_readcyclecounter() - k0; q[3] = kRounds; q[4] = ctl[slot].nbSeq;
    }
#endif
    wave_sync();         // the round stores are complete before the end record goes on top of one of them
    if (helper && hasJob && gl == 0) {
        DevBlockState* const bs = p.state + b;
        if (failU) {
            bs->seqErr = E_corruption; bs->seqValid = 0; bs->tailErr = 0; bs->regen = blk.litSize;
            bs->repOut[0] = kSlot0; bs->repOut[1] = kSlot1; bs->repOut[2] = kSlot2;
            out[0] = make_uint4(0, 0, 0, 0);
        } else {
            out[nValid] = make_uint4(0, outPos, litPos, 0);      // end record
            bs->seqErr = 0;
            bs->seqValid = nValid;
            // A stream that ran dry before the block's last sequence is not an error in the reference: its reads go on
            // answering from the last container (bitstream.pas:300-314).  Such blocks (damaged input only) are decoded
            // again, serially and to the letter, by k_sequences_dry.
            if (nValid < nbSeq) { uint32_t const at = atomicAdd(p.dryList, 1u); p.dryList[1 + at] = b; }
            // after the last sequence the reader must be at (or, through the last state update, past) the stream start (:1301);
            // the prefetching loop of 1.4.8 has no such check
            bool unfinished = nValid == nbSeq && (int)Qend > (int)Qstart;
            if (unfinished && prefetching_loop_148(p.frames[blk.frameIdx].longWindow, nbSeq, tOF, (ctl[slot].logs >> 8) & 255)) unfinished = false;
            bs->tailErr = unfinished ? (uint32_t)E_corruption : 0u;
            bs->regen = (uint64_t)outPos + (litPos <= blk.litSize ? blk.litSize - litPos : 0);
            bs->repOut[0] = rep.a; bs->repOut[1] = rep.b; bs->repOut[2] = rep.c;
        }
    }
}

// ---------------------------------------------------------------- stage 1b: sequence streams that run dry
// BIT_DStream_t to the letter (bitstream.pas:40-51, :206-398): a 64-bit container, a count of consumed bits that may pass
// 64, and reads that then keep answering from the container in hand (the two read forms wrap differently).
struct DStreamEmu {
    const uint8_t* start; const uint8_t* ptr; const uint8_t* limit;
    uint64_t c; uint32_t consumed;
    DEVI bool init(const uint8_t* src, uint32_t size)      // false: empty stream or no end mark
    {
        start = src; limit = src + 8; c = 0; consumed = 0; ptr = src;
        if (size < 1) return false;
        uint32_t const last = src[size - 1];
        if (size >= 8) { ptr = src + size - 8; c = ld_u64(ptr); consumed = last ? 8 - hb32(last) : 0; }
        else {
            for (uint32_t i = 0; i < size; i++) c |= (uint64_t)src[i] << (8 * i);
            consumed = (last ? 8 - hb32(last) : 0) + (8 - size) * 8;
        }
        return last != 0;
    }
    DEVI uint32_t look(uint32_t n) const { uint32_t const st = 64u - consumed - n; return (uint32_t)((c >> (st & 63)) & ((1ull << n) - 1)); }      // BIT_lookBits
    DEVI uint32_t lookFast(uint32_t n) const { return (uint32_t)((c << (consumed & 63)) >> ((64 - n) & 63)); }                                // BIT_lookBitsFast, n >= 1
    DEVI uint32_t read(uint32_t n) { uint32_t const v = look(n); consumed += n; return v; }
    DEVI uint32_t readFast(uint32_t n) { uint32_t const v = lookFast(n); consumed += n; return v; }
    // BIT_reloadDStream: 0 unfinished, 1 end of buffer, 2 completed, 3 overflow
    DEVI uint32_t reload()
    {
        if (consumed > 64) return 3;
        if (ptr >= limit) { ptr -= consumed >> 3; consumed &= 7; c = ld_u64(ptr); return 0; }
        if (ptr == start) return consumed < 64 ? 1u : 2u;
        uint32_t nb = consumed >> 3, res = 0;
        if (ptr - nb < start) { nb = (uint32_t)(ptr - start); res = 1; }
        ptr -= nb; consumed -= nb * 8; c = ld_u64(ptr);
        return res;
    }
};

// One table of a block, built by one lane: from the block's own description at payload offset `off` (advanced past
// it), from the dictionary, or from the earlier block that defined it.  false on an invalid description.
template <typename NormPtr, typename NextPtr>
DEVI bool serial_seq_table(const BatchParams& p, const DevBlock& blk, int t, uint32_t* cells, NormPtr norm, NextPtr symNext, uint32_t& off, uint32_t& log)
{
    uint32_t const type = t == T_LL ? blk.llType : t == T_OF ? blk.ofType : blk.mlType;
    int32_t const srcB = t == T_LL ? blk.llSrc : t == T_OF ? blk.ofSrc : blk.mlSrc;
    const uint8_t* const payload = p.src + blk.srcOff;
    if (type != SET_REPEAT) {
        if (off > blk.cSize) return false;
        uint32_t const h = dev_build_from_desc(t, type, payload + off, blk.cSize - off, cells, norm, symNext, log);
        if (h == kFail) return false;
        off += h;
        return true;
    }
    if (srcB == SRC_DICT) {
        log = t == T_LL ? p.dict->llLog : t == T_OF ? p.dict->ofLog : p.dict->mlLog;
        const uint32_t* const dc = t == T_LL ? p.dict->ll : t == T_OF ? p.dict->of : p.dict->ml;
        for (uint32_t u = 0; u < (1u << log); u++) cells[u] = dc[u];
        return true;
    }
    if (srcB == SRC_NONE) return false;
    DevBlock const sb = p.blocks[srcB];
    const uint8_t* const sp = p.src + sb.srcOff;
    uint32_t o = sb.seqHdrOff;
    for (int u = 0; u < t; u++) {      // (T_LL, T_OF, T_ML are the order of the descriptions)
        uint32_t const sz = dev_desc_size(u, u == T_LL ? sb.llType : sb.ofType, sp, o, sb.cSize, norm);
        if (sz == kFail) return false;
        o += sz;
    }
    if (o > sb.cSize) return false;
    uint32_t const stype = t == T_LL ? sb.llType : t == T_OF ? sb.ofType : sb.mlType;
    return dev_build_from_desc(t, stype, sp + o, sb.cSize - o, cells, norm, symNext, log) != kFail;
}

// Blocks listed by k_sequences: decoded again from the first sequence by one lane, exactly as
// ZSTD_decompressSequences_body reads them (zstd_decompress_block.pas:1232-1320, ZSTD_decodeSequence :1083-1230), the
// records of the fast path overwritten.  Input that takes this path is damaged; its cost does not matter.
__global__ __launch_bounds__(64) void k_sequences_dry(BatchParams p)
{
    __shared__ uint32_t tLL[512], tML[512], tOF[256];
    __shared__ int16_t norm[64];
    __shared__ uint16_t symNext[64];
    if (threadIdx.x != 0) return;
    uint32_t const count = p.dryList[0];
    for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) {
        uint32_t const b = p.dryList[1 + i];
        DevBlock const blk = p.blocks[b];
        DevBlockState* const st = p.state + b;
        uint4* const out = (uint4*)p.seqBuf + blk.seqBufOff;
        uint32_t off = blk.seqHdrOff, logLL = 0, logOF = 0, logML = 0;
        bool ok = serial_seq_table(p, blk, T_LL, tLL, norm, symNext, off, logLL) && serial_seq_table(p, blk, T_OF, tOF, norm, symNext, off, logOF) &&
                  serial_seq_table(p, blk, T_ML, tML, norm, symNext, off, logML) && off < blk.cSize;
        DStreamEmu bs;
        ok = ok && bs.init(p.src + blk.srcOff + off, blk.cSize - off);
        if (!ok) {      // (not reached: the fast path built the same tables and opened the same stream)
            st->seqErr = E_corruption; st->seqValid = 0; st->tailErr = 0; st->regen = blk.litSize;
            st->repOut[0] = kSlot0; st->repOut[1] = kSlot1; st->repOut[2] = kSlot2;
            out[0] = make_uint4(0, 0, 0, 0);
            continue;
        }
        // ZSTD_initFseState x3: LL, OF, ML (:1266-1268)
        uint32_t sLL = bs.read(logLL); bs.reload();
        uint32_t sOF = bs.read(logOF); bs.reload();
        uint32_t sML = bs.read(logML); bs.reload();
        Rep3 rep = {kSlot0, kSlot1, kSlot2};
        uint32_t outPos = 0, litPos = 0;
        // the prefetching loop looks at the stream before it decodes a sequence and stops when it has overflowed
        // (:1347, :1376); firstDry = the first sequence that finds it so
        bool const prefetching = prefetching_loop_148(p.frames[blk.frameIdx].longWindow, blk.nbSeq, tOF, logOF);
        uint32_t firstDry = blk.nbSeq;
        for (uint32_t n = 0; n < blk.nbSeq; n++) {
            if (prefetching && firstDry == blk.nbSeq && bs.reload() == 3) firstDry = n;
            uint32_t const cLL = tLL[sLL], cML = tML[sML], cOF = tOF[sOF];
            uint32_t const ofBits = cell_addBits(T_OF, cOF), mlBits = cell_addBits(T_ML, cML), llBits = cell_addBits(T_LL, cLL);
            uint32_t const llCode = cell_llCode(cLL), mlCode = cell_mlCode(cML);
            uint32_t ta;
            bool isReal = ofBits > 1;
            uint32_t idx = 0;
            if (isReal) {
                uint32_t const real = ((1u << ofBits) - 3u) + bs.readFast(ofBits);
                ta = real > 0x7FFFFFFFu ? 0x7FFFFFFFu : real;
            } else {
                idx = (ofBits ? 1u + bs.readFast(1) : 0u) + (llCode == 0 ? 1u : 0u);
                ta = idx == 0 ? kSlot0 : idx == 1 ? kSlot1 : idx == 2 ? kSlot2 : (kSlot0 | 1u);
            }
            Rep3 nx;
            nx.a = rep_subst(ta, rep.a, rep.b, rep.c);
            nx.b = (isReal || idx >= 1) ? rep.a : rep.b;
            nx.c = (isReal || idx >= 2) ? rep.b : rep.c;
            rep = nx;
            uint32_t ml = c_ML_base[mlCode < 53 ? mlCode : 0];
            if (mlBits) ml += bs.readFast(mlBits);
            if (ofBits + mlBits + llBits >= 31) bs.reload();       // :1171-1174
            uint32_t ll = c_LL_base[llCode < 36 ? llCode : 0];
            if (llBits) ll += bs.readFast(llBits);
            // state update: LL, ML, OF (:1220-1225)
            sLL = (cLL & 0x1FFu) + bs.read(cell_nbBits(cLL));
            sML = (cML & 0x1FFu) + bs.read(cell_nbBits(cML));
            sOF = (cOF & 0x1FFu) + bs.read(cell_nbBits(cOF));
            bs.reload();
            out[n] = make_uint4(rep.a, outPos, litPos, ll);
            uint64_t const no = (uint64_t)outPos + ll + ml, nl = (uint64_t)litPos + ll;
            outPos = no > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)no;
            litPos = nl > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)nl;
        }
        out[blk.nbSeq] = make_uint4(0, outPos, litPos, 0);
        st->seqErr = 0;
        if (prefetching) {
            // Executed before the loop gave up: the sequences up to four before the one that found the stream dry (the
            // loop decodes four ahead of the one it executes, :1341-1400); an error of theirs comes first, else the
            // missing sequences make the block corruption_detected.  No check of what is left of the stream.
            st->seqValid = firstDry == blk.nbSeq ? blk.nbSeq : firstDry >= 4 ? firstDry - 4 : 0u;
            st->tailErr = 0;
        } else {
            st->seqValid = blk.nbSeq;
            st->tailErr = bs.reload() < 2 ? (uint32_t)E_corruption : 0u;       // (:1301)
        }
        st->regen = (uint64_t)outPos + (litPos <= blk.litSize ? blk.litSize - litPos : 0);
        st->repOut[0] = rep.a; st->repOut[1] = rep.b; st->repOut[2] = rep.c;
    }
}

// ---------------------------------------------------------------- stage 2: per-frame layout
__global__ void k_layout(BatchParams p)
{
    uint32_t const fi = blockIdx.x * blockDim.x + threadIdx.x;
    if (fi >= p.nFrames) return;
    DevFrame const f = p.frames[fi];
    DevFrameStatus* const fs = p.status + fi;
    fs->xxh64Low32 = 0; fs->regenSize = 0;
    fs->errorCode = f.hostError;
    if (f.hostError) return;
    if (batch_overflow(p)) { fs->errorCode = E_memory_allocation; return; }     // the batch exceeds its fixed scratch
    if (f.useDict && p.dict->status) { fs->errorCode = p.dict->status; return; }
    uint32_t rep[3] = {1, 4, 8};   // repStartValue (zstd_internal.pas:13)
    if (f.useDict && p.dict->hasEntropy) { rep[0] = p.dict->rep[0]; rep[1] = p.dict->rep[1]; rep[2] = p.dict->rep[2]; }
    uint64_t pos = 0;
    for (uint32_t b = f.firstBlock; b < f.firstBlock + f.nBlocks; b++) {
        DevBlock const blk = p.blocks[b];
        DevBlockState* const st = p.state + b;
        st->dstOff = pos;
        st->repIn[0] = rep[0]; st->repIn[1] = rep[1]; st->repIn[2] = rep[2];
        uint64_t size;
        if (blk.blockType == BT_STOP) break;
        if (blk.blockType != BT_COMPRESSED) size = blk.origSize;
        else if (blk.nbSeq == 0 || blk.hostLitErr || blk.hostSeqErr) size = blk.litSize;
        else {
            size = st->regen;
            uint32_t const n0 = rep_resolve(st->repOut[0], rep), n1 = rep_resolve(st->repOut[1], rep), n2 = rep_resolve(st->repOut[2], rep);
            rep[0] = n0; rep[1] = n1; rep[2] = n2;
        }
        st->regen = size;
        pos += size;
        if (pos > (1ull << 62)) pos = 1ull << 62;
    }
}

// ---------------------------------------------------------------- stage 3: execute
DEVI void coop_copy(uint8_t* dst, const uint8_t* src, uint64_t n, uint32_t lane)
{
    for (uint64_t i = lane; i < n; i += 64) dst[i] = src[i];
}

// Context of one batch of <= 64 sequences for byte resolution.  Positions are relative to the first
// output byte of the batch; sStart/sLL/sOff/sLit describe the batch's sequences in LDS (sStart padded
// to 65 entries with the batch size).
struct ExecCtx {
    const uint32_t *sStart, *sLL, *sOff, *sLit;
    const uint8_t* litBase;
    const uint8_t* fdst;        // frame output
    const uint8_t* dictEnd;     // one past the dictionary content
    int64_t batchAbs;           // frame-relative position of batch byte 0
    bool litRle;
};

// last sequence whose start <= q
DEVI uint32_t owner_of(const ExecCtx& c, uint32_t q)
{
    uint32_t lo = 0;
#pragma unroll
    for (int step = 32; step >= 1; step >>= 1) { uint32_t const mid = lo + step; lo = (c.sStart[mid] <= q) ? mid : lo; }
    return lo;
}

// Where the byte that ends up at batch position k comes from; `own` is the sequence producing k.
// A position before the batch is already in memory (frame output, or dictionary when it reaches
// before the frame).  A position inside the batch is either a literal, or a match byte whose own
// source is followed in turn -- each hop lands in an earlier sequence, so the walk ends; overlapping
// matches (offset < matchLength) fold with a modulo.  This replaces the byte-serial
// ZSTD_overlapCopy8 / ZSTD_safecopy (zstd_decompress_block.pas:793-885).  Wave-uniform loop.
// `litAt` receives the literal position (block-relative) when the byte is a literal, else 0xFFFFFFFF.
DEVI const uint8_t* resolve_source(const ExecCtx& c, uint32_t k, uint32_t own, bool active, uint32_t& litAt)
{
    const uint8_t* ptr = c.litBase;
    litAt = 0xFFFFFFFFu;
    uint32_t q = k;
    bool pending = active;
    for (int hop = 0; hop < 66 && __any(pending); hop++) {
        uint32_t const st = c.sStart[own];
        uint32_t const r = q - st, sll = c.sLL[own];
        const uint8_t* cand;
        bool resolved;
        uint32_t candLit = 0xFFFFFFFFu;
        if (r < sll) { candLit = c.sLit[own] + r; cand = c.litRle ? c.litBase : c.litBase + candLit; resolved = true; }
        else {
            uint32_t m = r - sll;
            uint32_t const so = c.sOff[own];
            if (m >= so) m %= so;                       // overlapping match: the pattern of `so` bytes repeats
            int32_t const nq = (int32_t)(st + sll + m) - (int32_t)so;   // so <= 2^31-1 and positions < 2^31: no wrap
            int64_t const abs = c.batchAbs + nq;
            cand = abs >= 0 ? c.fdst + abs : c.dictEnd + abs;
            resolved = nq < 0;
            if (pending && !resolved) { q = (uint32_t)nq; own = owner_of(c, q); }
        }
        if (pending && resolved) { ptr = cand; litAt = candLit; pending = false; }
    }
    return ptr;
}

// One run (a sequence's literals, or a match whose source lies wholly before the batch and does not
// overlap its destination) copied by one lane with exact-size unaligned accesses: up to two 16-byte
// chunks and the 8/4/2/1-byte pieces of the remainder.  load() issues every load of the run, store()
// every store, so a wave can issue all loads of a batch before its first store (a wave's loads and
// stores retire in issue order).  Runs of 48 bytes or more keep `more` 16-byte chunks for a
// cooperative loop.
struct Run {
    uint4 a, b;
    uint2 c8;
    uint32_t c4;
    uint16_t c2;
    uint8_t c1;
    DEVI void load(const uint8_t* src, uint32_t len, bool splat)
    {
        uint32_t const n16 = len >> 4;
        if (splat) {     // RLE literals: every byte is src[0]
            uint32_t const v = 0x01010101u * src[0];
            a = b = make_uint4(v, v, v, v); c8 = make_uint2(v, v); c4 = v; c2 = (uint16_t)v; c1 = (uint8_t)v;
            return;
        }
        if (n16 >= 1) __builtin_memcpy(&a, src, 16);
        if (n16 >= 2) __builtin_memcpy(&b, src + 16, 16);
        uint32_t po = n16 << 4;
        if (len & 8) { __builtin_memcpy(&c8, src + po, 8); po += 8; }
        if (len & 4) { __builtin_memcpy(&c4, src + po, 4); po += 4; }
        if (len & 2) { __builtin_memcpy(&c2, src + po, 2); po += 2; }
        if (len & 1) c1 = src[po];
    }
    DEVI void store(uint8_t* dst, uint32_t len) const
    {
        uint32_t const n16 = len >> 4;
        if (n16 >= 1) __builtin_memcpy(dst, &a, 16);
        if (n16 >= 2) __builtin_memcpy(dst + 16, &b, 16);
        uint32_t po = n16 << 4;
        if (len & 8) { __builtin_memcpy(dst + po, &c8, 8); po += 8; }
        if (len & 4) { __builtin_memcpy(dst + po, &c4, 4); po += 4; }
        if (len & 2) { __builtin_memcpy(dst + po, &c2, 2); po += 2; }
        if (len & 1) dst[po] = c1;
    }
};

// chunks 2 .. n16-1 of every run longer than 47 bytes, copied by the whole wave (16 bytes per lane)
DEVI void copy_long_runs(uint8_t* dst, const uint8_t* src, uint32_t len, bool splat, uint32_t lane)
{
    unsigned long long longs = __ballot((len >> 4) > 2);
    while (longs) {
        uint32_t const who = (uint32_t)__builtin_ctzll(longs);
        longs &= longs - 1;
        uint32_t const n16 = __shfl(len, who) >> 4;
        uint8_t* const d = (uint8_t*)(uintptr_t)__shfl((unsigned long long)(uintptr_t)dst, who);
        const uint8_t* const sp = (const uint8_t*)(uintptr_t)__shfl((unsigned long long)(uintptr_t)src, who);
        bool const sv = __shfl((int)splat, who) != 0;
        uint32_t const v = sv ? 0x01010101u * sp[0] : 0;
        for (uint32_t i = 2 + lane; i < n16; i += 64) {
            uint4 x = make_uint4(v, v, v, v);
            if (!sv) __builtin_memcpy(&x, sp + 16 * (size_t)i, 16);
            __builtin_memcpy(d + 16 * (size_t)i, &x, 16);
        }
    }
}

// One batch copied straight from memory to memory: the path for batches with long runs (literal runs or matches
// over 32 bytes, or more output than the LDS stage holds).  Kept out of line: it is the rare path.
__attribute__((noinline)) static __device__ void exec_batch_direct(const uint32_t* sStart, const uint32_t* sLL, const uint32_t* sOff, const uint32_t* sLit,
                                                                   const uint32_t* sSlowEnd, const uint8_t* litBase, const uint8_t* fdst, const uint8_t* dictEnd,
                                                                   int64_t batchAbs, bool litRle, uint8_t* myDst, const uint8_t* litSrc, const uint8_t* matSrc,
                                                                   uint32_t ll, uint32_t fastLen, uint32_t slowTotal, uint8_t* batchDst, uint32_t lane)
{
    ExecCtx c;      // (arguments by value: a context passed by reference would live in scratch memory on the caller's hot path too)
    c.sStart = sStart; c.sLL = sLL; c.sOff = sOff; c.sLit = sLit;
    c.litBase = litBase; c.fdst = fdst; c.dictEnd = dictEnd;
    c.batchAbs = batchAbs; c.litRle = litRle;
    // ---- all loads of the batch ...
    Run rl, rm;
    rl.load(litSrc, ll, c.litRle);
    rm.load(matSrc, fastLen, false);
    // first pass of the slow bytes: byte j of the concatenated slow matches -> owner by search over the prefix
    uint32_t sk = 0, la;
    bool const sAct = lane < slowTotal;
    const uint8_t* sSrc = c.litBase;
    if (slowTotal) {
        uint32_t own = 0;
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1) own += (sSlowEnd[own + step - 1] <= lane) ? (uint32_t)step : 0u;
        own = sAct ? own : 0;
        uint32_t const mlo = c.sStart[own + 1] - c.sStart[own] - c.sLL[own];
        sk = c.sStart[own] + c.sLL[own] + (lane - (sSlowEnd[own] - mlo));
        sSrc = resolve_source(c, sAct ? sk : 0, own, sAct, la);
    }
    uint8_t sv = 0;
    if (slowTotal && sAct) sv = *sSrc;
    // ---- ... then all stores
    rl.store(myDst, ll);
    rm.store(myDst + ll, fastLen);
    if (slowTotal && sAct) batchDst[sk] = sv;
    copy_long_runs(myDst, litSrc, ll, c.litRle, lane);
    copy_long_runs(myDst + ll, matSrc, fastLen, false, lane);
    // remaining slow bytes, 64 per pass
    for (uint32_t j0 = 64; j0 < slowTotal; j0 += 64) {
        uint32_t const j = j0 + lane;
        bool const act = j < slowTotal;
        uint32_t own = 0;
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1) own += (sSlowEnd[own + step - 1] <= j) ? (uint32_t)step : 0u;
        own = act ? own : 0;
        uint32_t const mlo = c.sStart[own + 1] - c.sStart[own] - c.sLL[own];
        uint32_t const k = c.sStart[own] + c.sLL[own] + (j - (sSlowEnd[own] - mlo));
        const uint8_t* const sp = resolve_source(c, act ? k : 0, own, act, la);
        if (act) batchDst[k] = *sp;
    }
}

// Two runs of 0..32 bytes each copied inside LDS (unaligned).  A run is two, possibly overlapping, accesses of one
// size: 16 bytes at the start and 16 at the end for 16..32, likewise 8 / 4 for shorter runs, three single bytes for
// runs of 1..3 -- an exact cover.  The two runs go through each size class together.
template <typename T>
DEVI void lds_copy_class(uint8_t* dstA, const uint8_t* srcA, uint32_t lenA, uint8_t* dstB, const uint8_t* srcB, uint32_t lenB, uint32_t lo, uint32_t hi)
{
    bool const inA = lenA >= lo && lenA < hi, inB = lenB >= lo && lenB < hi;
    T a0, a1, b0, b1;
    if (inA) { __builtin_memcpy(&a0, srcA, sizeof(T)); __builtin_memcpy(&a1, srcA + (lenA - sizeof(T)), sizeof(T)); }
    if (inB) { __builtin_memcpy(&b0, srcB, sizeof(T)); __builtin_memcpy(&b1, srcB + (lenB - sizeof(T)), sizeof(T)); }
    if (inA) { __builtin_memcpy(dstA, &a0, sizeof(T)); __builtin_memcpy(dstA + (lenA - sizeof(T)), &a1, sizeof(T)); }
    if (inB) { __builtin_memcpy(dstB, &b0, sizeof(T)); __builtin_memcpy(dstB + (lenB - sizeof(T)), &b1, sizeof(T)); }
}
DEVI void lds_copy_runs(uint8_t* dstA, const uint8_t* srcA, uint32_t lenA, uint8_t* dstB, const uint8_t* srcB, uint32_t lenB)
{
    lds_copy_class<uint4>(dstA, srcA, lenA, dstB, srcB, lenB, 16, 33);
    lds_copy_class<uint2>(dstA, srcA, lenA, dstB, srcB, lenB, 8, 16);
    lds_copy_class<uint32_t>(dstA, srcA, lenA, dstB, srcB, lenB, 4, 8);
    // runs of 1..3 bytes: first, middle and last byte (for 1 and 2 bytes some of them coincide)
    {
        bool const inA = lenA >= 1 && lenA < 4, inB = lenB >= 1 && lenB < 4;
        uint8_t a0 = 0, a1 = 0, a2 = 0, b0 = 0, b1 = 0, b2 = 0;
        if (inA) { a0 = srcA[0]; a1 = srcA[lenA >> 1]; a2 = srcA[lenA - 1]; }
        if (inB) { b0 = srcB[0]; b1 = srcB[lenB >> 1]; b2 = srcB[lenB - 1]; }
        if (inA) { dstA[0] = a0; dstA[lenA >> 1] = a1; dstA[lenA - 1] = a2; }
        if (inB) { dstB[0] = b0; dstB[lenB >> 1] = b1; dstB[lenB - 1] = b2; }
    }
}

// A value known to be the same in every lane, moved to a scalar register.
DEVI uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
DEVI uint64_t uni(uint64_t v) { return ((uint64_t)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v); }
DEVI uint32_t lane_value(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }

// Whether a frame is decoded by k_execute_wf (eight waves) or by k_execute (one): by eight waves when its blocks carry
// enough sequences on average for the waves to have batches to share.  Same answer in both kernels.  p.execMode
// forces one or the other (set by launch_execute).
DEVI bool frame_uses_wf(const BatchParams& p, const DevFrame& f)
{
    if (p.execMode) return p.execMode == 2;
    uint64_t seqs = 0;
    uint32_t const nb = f.nBlocks < 64 ? f.nBlocks : 64;       // (the first blocks tell)
    for (uint32_t b = f.firstBlock; b < f.firstBlock + nb; b++) seqs += p.blocks[b].nbSeq;
    return nb != 0 && seqs >= 1024ull * nb;
}

// (the three sizes can be set from the compiler command line for experiments: how the stage's time depends on its LDS)
#ifndef ZG_OUT_STAGE
#define ZG_OUT_STAGE 4096
#endif
#ifndef ZG_OUT_HIST
#define ZG_OUT_HIST 7168
#endif
#ifndef ZG_LIT_STAGE
#define ZG_LIT_STAGE 1024
#endif
constexpr uint32_t kOutStage = ZG_OUT_STAGE;     // output bytes of one batch assembled in LDS
constexpr uint32_t kOutHist = ZG_OUT_HIST;       // the output stage: a batch and the output of the batches before it
constexpr uint32_t kOutKeep = kOutHist - kOutStage;     // history kept when the stage is full (its tail moves to the front)
constexpr uint32_t kLitStage = ZG_LIT_STAGE;     // literal bytes of one batch staged in LDS
static_assert(kOutStage <= kOutHist && kOutStage % 16 == 0 && kOutHist % 16 == 0 && kLitStage % 16 == 0 && kLitStage <= 1024, "stage sizes");

// kMatch64: plain matches of up to 64 bytes (instead of 32) are assembled in LDS too, in a second step.  Worth its
// instructions where such matches turn up in most batches -- records compressed against a dictionary, whose matches into
// the dictionary run long -- and a loss elsewhere (the step is never taken, but its code costs the usual path registers):
// the launcher picks the variant by whether the batch has a dictionary.
template <bool kMatch64>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void k_execute(BatchParams p)
{
    // The output stage holds the batch being assembled and, before it, what the batches before it produced (as long as
    // they were assembled here and the next one still fits): a match whose source lies in that history is copied
    // inside LDS and costs no memory access.
    __shared__ __attribute__((aligned(16))) uint8_t outStage[kOutHist];
    __shared__ __attribute__((aligned(16))) uint8_t litStage[kLitStage];
    __shared__ __attribute__((aligned(16))) uint8_t matStage[64 * 32];      // 32 source bytes per lane
    // the sequence table of a batch that goes from memory to memory (it uses none of the stages)
    uint32_t* const sStart = (uint32_t*)matStage;
    uint32_t* const sLL = sStart + 65;
    uint32_t* const sOff = sLL + 64;
    uint32_t* const sLit = sOff + 64;
    uint32_t* const sSlowEnd = sLit + 64;
    static_assert((65 + 4 * 64) * 4 <= 64 * 32, "the sequence table fits in the match stage");
    uint32_t const lane = threadIdx.x;
    uint32_t const fi = blockIdx.x;
#ifdef ZG_X_LDSPAD    // experiment: LDS the kernel does not use, to hold fewer waves on a compute unit
    __shared__ uint32_t ldsPad[ZG_X_LDSPAD / 4];
    asm volatile("" :: "v"((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint32_t*)&ldsPad[lane]));
#endif
    DevFrame const f = p.frames[fi];
    DevFrameStatus* const fs = p.status + fi;
    if (fs->errorCode) return;   // host error or unusable dictionary (set by k_layout)
    if (frame_uses_wf(p, f)) return;        // k_execute_wf's
    uint8_t* const fdst = p.dst + f.dstOff;
    uint64_t const cap = f.dstCapacity;
    uint32_t const dictN = f.useDict ? p.dictContentSize : 0;
    const uint8_t* const dictEnd = p.dictContent + dictN;
    uint32_t ferr = 0;
    uint64_t produced = 0;

    for (uint32_t b = f.firstBlock; b < f.firstBlock + f.nBlocks && !ferr; b++) {
        // block descriptor and state are the same for every lane: keep what the loops below use in scalar registers
        const DevBlock* const bp = p.blocks + b;
        const DevBlockState* const sp_ = p.state + b;
        uint32_t const blockType = uni((uint32_t)bp->blockType), origSize = uni(bp->origSize);
        uint64_t const blkSrcOff = uni(bp->srcOff);
        uint32_t const hostLitErr = uni((uint32_t)bp->hostLitErr), hostSeqErr = uni((uint32_t)bp->hostSeqErr);
        uint32_t const litType = uni((uint32_t)bp->litType), nbSeq = uni(bp->nbSeq);
        uint64_t const bo = produced;           // block start inside the frame (== the layout stage's dstOff)
        uint64_t const room = cap - bo;
        if (blockType == BT_STOP) { ferr = hostLitErr; break; }
        if (blockType == BT_RAW) {
            if (origSize > room) { ferr = E_dstSize_tooSmall; break; }
            coop_copy(fdst + bo, p.src + blkSrcOff, origSize, lane);
            produced += origSize;
            continue;
        }
        if (blockType == BT_RLE) {
            if (origSize > room) { ferr = E_dstSize_tooSmall; break; }
            uint8_t const v = p.src[blkSrcOff];
            for (uint64_t i = lane; i < origSize; i += 64) fdst[bo + i] = v;
            produced += origSize;
            continue;
        }
        // compressed block: error precedence follows ZSTD_decompressBlock_internal (zstd_decompress_block.pas:1491-1542)
        if (hostLitErr) { ferr = hostLitErr; break; }
        bool const hufLits = litType >= SET_COMPRESSED;
        uint32_t const litErr = uni(sp_->litErr), seqErr = uni(sp_->seqErr), tailErr = uni(sp_->tailErr);
        if (hufLits && litErr) { ferr = litErr; break; }
        if (hostSeqErr) { ferr = hostSeqErr; break; }
        if (nbSeq && seqErr) { ferr = seqErr; break; }

        const uint8_t* const litBase = hufLits ? p.litBuf + uni(bp->litBufOff) : p.src + blkSrcOff + uni((uint32_t)bp->litHdrSize);
        bool const litRle = (litType == SET_RLE);
        uint32_t const litSize = uni(bp->litSize);
        uint32_t const rleFill = (litRle && litSize) ? 0x01010101u * uni((uint32_t)litBase[0]) : 0u;
        uint32_t const nValid = nbSeq ? uni(sp_->seqValid) : 0;
        const uint4* const recs = (const uint4*)p.seqBuf + uni(bp->seqBufOff);   // nbSeq + 1 records
        uint32_t const rin0 = uni(sp_->repIn[0]), rin1 = uni(sp_->repIn[1]), rin2 = uni(sp_->repIn[2]);
        uint64_t pos = 0;       // bytes of this block already produced
        uint32_t litPos = 0;

        // One sequence per lane: record = (offset, output position, literal position, litLength), block-relative;
        // lanes past the batch's last sequence read the record after it, whose output position ends the batch.
        // The loop is software-pipelined: while the loads of a batch are in flight the plan of the next batch is
        // worked out from its records (requested one batch earlier), and the records of the batch after that are
        // requested.
        uint4 rcN = make_uint4(0, 0, 0, 0), rcB = rcN;        // records of the next batch (arrived) and of the one after it (in flight)
        uint2 endN = make_uint2(0, 0), endB = endN;             // {output position, literal position} of the record that ends those batches
        auto load_recs = [&](uint32_t s, uint4& rc, uint2& end) {     // records of the batch that starts at sequence s (past the end: the end record)
            uint32_t const sc = s < nValid ? s : nValid;
            uint32_t const n2 = nValid - sc < 64 ? nValid - sc : 64;
            // (read once: no need to keep them in the caches, which the match sources want)
            typedef uint32_t v4u __attribute__((ext_vector_type(4)));
            typedef uint32_t v2u __attribute__((ext_vector_type(2)));
            v4u const r4 = __builtin_nontemporal_load((const v4u*)(recs + sc + (lane < n2 ? lane : n2)));
            v2u const e2 = __builtin_nontemporal_load((const v2u*)((const uint32_t*)(recs + sc + n2) + 1));
            rc = make_uint4(r4.x, r4.y, r4.z, r4.w);
            end = make_uint2(e2.x, e2.y);
        };
        // What a batch needs, from its records alone.  Per lane: the sequence's start inside the batch, lengths, offset,
        // literal position, source position; the rest is the same in every lane.
        struct Plan {
            uint32_t start, ll, lstart, off, ml, gcnt, gEnd;
            int64_t absS;
            bool have, plain, fastS, nearS;
            uint32_t total, lit0, sumLL, base0, litEnd, err;
            uint32_t hp;        // where the batch starts in the output stage = bytes of history in front of it
            uint32_t shift;
            bool staged;
        };
        // prev: the plan of the batch before (its place in the stage decides how much history this one finds)
        auto make_plan = [&](uint4 const rc, uint2 const endV, uint32_t const s, bool const prevStaged, uint32_t const prevEnd) -> Plan {
            Plan q;
            uint32_t const n = s < nValid ? (nValid - s < 64 ? nValid - s : 64) : 0;
            q.have = lane < n;
            uint32_t const endY = uni(endV.x), endZ = uni(endV.y);
            uint32_t const below = wave_next(rc.y);
            uint32_t const nstart = lane == 63 ? endY : (q.have ? below : rc.y);
            uint32_t const bstart = rc.y;
            q.lstart = rc.z;
            q.ll = q.have ? rc.w : 0;
            // (straight-line on purpose, selects and bitwise logic instead of branches: a branch costs this single wave more
            // than the few instructions it would skip)
            uint32_t const offR = rep_subst(rc.x, rin0, rin1, rin2);
            q.off = q.have ? offR : 1;
            // per-sequence checks in the order of ZSTD_execSequenceEnd / ZSTD_execSequence (:896-1008)
            bool const e1 = nstart > room;
            bool const e2 = (uint64_t)q.lstart + q.ll > litSize;
            bool const e3 = (uint64_t)q.off > bo + bstart + q.ll + dictN;
            uint32_t const code = !q.have ? 0u : e1 ? (uint32_t)E_dstSize_tooSmall : (e2 | e3) ? (uint32_t)E_corruption : 0u;
            unsigned long long const bad = __ballot(code != 0);
            q.err = bad ? uni((uint32_t)__shfl(code, __builtin_ctzll(bad))) : 0u;
            q.ml = nstart - bstart - q.ll;
            q.base0 = lane_value(bstart, 0);           // block-relative start of the batch
            q.start = bstart - q.base0;
            q.total = lane_value(nstart, 63) - q.base0;
            q.litEnd = lane_value(q.lstart + q.ll, 63);
            int64_t const batchAbs = (int64_t)(bo + q.base0);
            // A match is "plain" when its source lies wholly before the batch, on one side of the dictionary
            // boundary, and does not overlap its destination: a copy from memory.  The others (about 1 match in 12 on
            // text) depend on bytes of the batch itself, or repeat a pattern.
            int64_t const ms = (int64_t)(q.start + q.ll) - (int64_t)q.off;          // batch-relative source start
            q.absS = batchAbs + ms;
            q.plain = q.have & (q.off >= q.ml) & (ms + (int64_t)q.ml <= 0) & ((q.absS >= 0) | (q.absS + (int64_t)q.ml <= 0));
            // Batches of short runs (the usual case) are assembled in LDS; there a plain match is fetched as 16 or 32
            // bytes whatever its length, which must stay inside the frame's output area or the dictionary.
            q.lit0 = lane_value(q.lstart, 0);
            q.sumLL = endZ - q.lit0;          // the literals of a batch are contiguous
            // (16 bytes at a time: up to 32 with the batch's other loads, the rest of a match of up to 64 bytes in a second step)
            int64_t const fetchEnd = q.absS + ((kMatch64 && q.ml > 48) ? 64 : (kMatch64 && q.ml > 32) ? 48 : q.ml > 16 ? 32 : 16);
            bool const fetchable = q.plain & (q.absS >= 0 ? fetchEnd <= (int64_t)cap : fetchEnd <= 0);
            uint32_t const slowS = (q.have & !fetchable) ? q.ml : 0;
            // of a match handled in LDS, the leading bytes whose source lies before the batch come from memory, one per lane
            int64_t const lead = -ms < (int64_t)(q.ml < q.off ? q.ml : q.off) ? -ms : (int64_t)(q.ml < q.off ? q.ml : q.off);
            q.gcnt = ((slowS != 0) & (lead > 0)) ? (uint32_t)lead : 0u;
            q.gEnd = wave_incl_scan(q.gcnt);
            bool const longRun = __any((q.ll > 32) | (q.plain & (q.ml > (kMatch64 ? 64u : 32u))));
            q.staged = (q.total <= kOutStage) & (q.sumLL <= kLitStage) & (lane_value(q.gEnd, 63) <= 64) & !longRun;
            // the batch goes right behind the previous one; when the stage is full the last kOutKeep bytes of history move to
            // the front first (`shift` = how far)
            q.hp = (q.staged && prevStaged) ? (prevEnd + q.total <= kOutHist ? prevEnd : (prevEnd < kOutKeep ? prevEnd : kOutKeep)) : 0u;
            q.shift = (q.staged && prevStaged) ? prevEnd - q.hp : 0u;
            q.nearS = fetchable & ((uint64_t)(-ms) <= q.hp);          // source inside the history: no fetch
            q.fastS = fetchable & !q.nearS;
            return q;
        };
        Plan pl = {};
        if (nValid) {
            load_recs(0, rcN, endN);
            pl = make_plan(rcN, endN, 0, false, 0);
            load_recs(64, rcN, endN);
        }
        for (uint32_t s0 = 0; s0 < nValid; s0 += 64) {
#ifdef ZG_STAMPS
            uint64_t const x0 = __builtin_readcyclecounter();
#endif
            if (pl.err) { ferr = pl.err; break; }
            load_recs(s0 + 128, rcB, endB);      // ahead of the batch's own loads: the wait for those covers it
            bool const have = pl.have, plain = pl.plain, fastS = pl.fastS, nearS = pl.nearS, staged = pl.staged;
            uint32_t const hp = pl.hp;
            uint8_t* const outB = outStage + hp;     // the batch's place in the output stage
            uint32_t const start = pl.start, ll = pl.ll, lstart = pl.lstart, off = pl.off, ml = pl.ml, gcnt = pl.gcnt, gEnd = pl.gEnd;
            uint32_t const total = pl.total, lit0 = pl.lit0, sumLL = pl.sumLL, base0 = pl.base0;
            int64_t const absS = pl.absS;
            int64_t const batchAbs = (int64_t)(bo + base0);
            uint32_t const slowS = (have && !fastS && !nearS) ? ml : 0;
#ifdef ZG_STAMPS
            unsigned long long const slowMask = __ballot(slowS != 0);
#endif
            const uint8_t* const matSrc = absS >= 0 ? fdst + absS : dictEnd + absS;
#ifdef ZG_STAMPS
            uint64_t const x1 = __builtin_readcyclecounter();
            uint64_t x2 = x1, x3 = x1, x4 = x1, xa = x1, xb = x1;
#endif
            // ---- all global loads of a batch assembled in LDS: one coalesced literal load, one or two 16-byte loads per
            // plain match, single bytes for source bytes before the batch of the other matches
            // Every lane issues every load, on a harmless address (the block's records) when it has nothing to fetch: a fixed
            // number of loads in flight lets the waits further down be counted instead of draining everything.
            uint32_t const fastLen = (staged && fastS) ? (ml < 32 ? ml : 32u) : 0;      // fetched with the batch's loads; the rest further down
            uint32_t const litFull = sumLL >> 4, litTail = sumLL & 15;
            const uint8_t* const safe = (const uint8_t*)recs;
            const uint8_t* const litG = litBase + lit0;
            bool const stagedLit = staged && !litRle;
            uint4 litV, mv0, mv1;
            __builtin_memcpy(&litV, (stagedLit && lane < litFull) ? litG + 16 * lane : safe, 16);
            uint8_t const litT = *((stagedLit && lane < litTail) ? litG + 16 * litFull + lane : safe);
#ifdef ZG_X_NOFAR     // experiment (wrong output): what the scattered match-source fetches cost
            __builtin_memcpy(&mv0, safe, 16);
            __builtin_memcpy(&mv1, safe, 16);
#else
            __builtin_memcpy(&mv0, fastLen ? matSrc : safe, 16);
            __builtin_memcpy(&mv1, fastLen > 16 ? matSrc + 16 : safe, 16);
#endif
            uint32_t gdst = 0xFFFFFFFFu;
            const uint8_t* gsrc = safe;
            {
                unsigned long long gm = staged ? __ballot(gcnt != 0) : 0ull;
                while (gm) {
                    int const w = __builtin_ctzll(gm);
                    gm &= gm - 1;
                    uint32_t const cnt = lane_value(gcnt, w), first = lane_value(gEnd, w) - cnt, d0 = lane_value(start + ll, w);
                    int64_t const absW = (int64_t)(((uint64_t)lane_value((uint32_t)((uint64_t)absS >> 32), w) << 32) | lane_value((uint32_t)absS, w));
                    uint32_t const j = lane - first;
                    if (j < cnt) {
                        int64_t const abs = absW + j;
                        gdst = d0 + j;
                        gsrc = abs >= 0 ? fdst + abs : dictEnd + abs;
                    }
                }
            }
            uint8_t const gv = *gsrc;
#ifdef ZG_STAMPS
            x2 = __builtin_readcyclecounter();
#endif
            // ---- while those are in flight: the next batch's plan
            Plan const plN = make_plan(rcN, endN, s0 + 64, staged, hp + total);
            if (staged) {
                // ---- literals and fetched match bytes into LDS, then every run into its place in the output stage
                lds_sync();       // the previous batch's reads of the stages are done
                if (pl.shift) {
                    // (source above destination, 1 KB per pass, every pass reads before it writes: what a pass overwrites has been read)
                    uint32_t const sh = pl.shift;
                    for (uint32_t o = 16 * lane; o < hp; o += 1024) {
                        uint4 x;
                        __builtin_memcpy(&x, outStage + sh + o, 16);       // (may run past the history's end into bytes nobody uses)
                        __builtin_memcpy(outStage + o, &x, 16);
                    }
                }
                if (litRle) { if (16 * lane < kLitStage) *(uint4*)(litStage + 16 * lane) = make_uint4(rleFill, rleFill, rleFill, rleFill); }
                else {
                    if (lane < litFull) *(uint4*)(litStage + 16 * lane) = litV;
                    if (lane < litTail) litStage[16 * litFull + lane] = litT;
                }
                if (fastLen) *(uint4*)(matStage + 32 * lane) = mv0;
                if (fastLen > 16) *(uint4*)(matStage + 32 * lane + 16) = mv1;
#ifdef ZG_STAMPS
                x3 = __builtin_readcyclecounter();
#endif
                lds_sync();
#ifndef ZG_X_NORUNS
                lds_copy_runs(outB + start, litStage + (litRle ? 0u : lstart - lit0), ll,
                              outB + start + ll, nearS ? outB + start + ll - off : matStage + 32 * lane, nearS ? (ml < 32 ? ml : 32u) : fastLen);
#endif
                if (gdst != 0xFFFFFFFFu) outB[gdst] = gv;
                // ---- plain matches of 33 to 64 bytes: bytes 32 and up, the same way (short dictionary-compressed records have
                // such a match in nearly every batch; without this step the whole batch would go from memory to memory)
                if (kMatch64) {
                    bool const longM = (fastS | nearS) & (ml > 32);
                    if (__any(longM)) {
                        uint32_t const rest = longM ? ml - 32 : 0u;
                        bool const far = longM & fastS;
                        uint4 w0, w1;
                        __builtin_memcpy(&w0, far ? matSrc + 32 : safe, 16);
                        __builtin_memcpy(&w1, (far && rest > 16) ? matSrc + 48 : safe, 16);
                        lds_sync();       // the first 32 bytes have left the match stage
                        if (far) { *(uint4*)(matStage + 32 * lane) = w0; if (rest > 16) *(uint4*)(matStage + 32 * lane + 16) = w1; }
                        lds_sync();
                        lds_copy_runs(outB + start + ll + 32, nearS ? outB + start + ll + 32 - off : matStage + 32 * lane, rest, outB, litStage, 0u);
                    }
                }
#ifdef ZG_STAMPS
                xa = __builtin_readcyclecounter();
#endif
                // ---- the other matches, one after the other in sequence order: everything before a match is in place
                // when its turn comes.  Byte j repeats pattern byte j mod offset; the pattern is the `offset` bytes before
                // the match, of which those before the batch were fetched to the head of the match itself.
                // (LDS operations of a wave execute in order: a read sees the writes issued before it.)
                {
                    unsigned long long sm = __ballot(slowS != 0);
#ifdef ZG_X_NOSLOW
                    sm = 0;
#endif
                    while (sm) {
                        int const w = __builtin_ctzll(sm);
                        sm &= sm - 1;
                        uint32_t const d0 = lane_value(start + ll, w), o = lane_value(off, w), m = lane_value(ml, w), g = lane_value(gcnt, w);
                        if (o < m) {
                            // (j mod o by a reciprocal: j, o < 2^12, the quotient estimate is off by one at most)
                            float const ro = __builtin_amdgcn_rcpf((float)o);
                            for (uint32_t j = g + lane; j < m; j += 64) {
                                uint32_t q = j - (uint32_t)((float)j * ro) * o;
                                q = (int32_t)q < 0 ? q + o : q;
                                q = q >= o ? q - o : q;
                                outB[d0 + j] = outB[q < g ? d0 + q : d0 - o + q];
                            }
                        } else {
                            for (uint32_t j = g + lane; j < m; j += 64) outB[d0 + j] = outB[d0 - o + j];
                        }
                    }
                }
#ifdef ZG_STAMPS
                xb = __builtin_readcyclecounter();
#endif
                lds_sync();
                // ---- the batch's output, contiguous: 16 bytes per lane, then the last bytes one per lane
                {
                    uint8_t* const d0 = fdst + bo + base0;
#ifdef ZG_X_NOSTORE
                    uint32_t const full = 0, tail = 0;
#else
                    uint32_t const full = total >> 4, tail = total & 15;
#endif
                    for (uint32_t u = lane; u < full; u += 64) {
                        uint4 x;
                        __builtin_memcpy(&x, outB + 16 * u, 16);
                        __builtin_memcpy(d0 + 16 * u, &x, 16);
                    }
                    if (lane < tail) d0[16 * full + lane] = outB[16 * full + lane];
                }
#ifdef ZG_STAMPS
                x4 = __builtin_readcyclecounter();
#endif
            } else {
                // Long runs: straight from memory to memory; the other matches byte by byte through the batch's
                // sequence table in LDS.
                uint32_t const fastLenD = plain ? ml : 0, slowLen = (have && !plain) ? ml : 0;
                uint32_t const slowEnd = wave_incl_scan(slowLen);
                uint32_t const slowTotal = lane_value(slowEnd, 63);
                lds_sync();       // previous LDS reads done
                sStart[lane] = start; sLL[lane] = ll; sOff[lane] = off; sLit[lane] = lstart; sSlowEnd[lane] = slowEnd;
                if (lane == 63) sStart[64] = total;
                lds_sync();
                exec_batch_direct(sStart, sLL, sOff, sLit, sSlowEnd, litBase, fdst, dictEnd, batchAbs, litRle, fdst + bo + base0 + start,
                                  litRle ? litBase : litBase + lstart, matSrc, ll, fastLenD, slowTotal, fdst + bo + base0, lane);
            }
            pos = (uint64_t)base0 + total;
            litPos = pl.litEnd;
#ifdef ZG_STAMPS
            if (p.dbg && lane == 0) {
                uint64_t const x5 = __builtin_readcyclecounter();
                uint64_t* const g = p.dbg + 8 * fi;
                g[0] += x1 - x0; g[1] += x3 - x1; g[2] += xa - x3; g[3] += xb - xa; g[4] += x4 - xb; g[5] += 1; g[6] += uni((uint32_t)__builtin_popcountll(slowMask)); g[7] += x5 - x4;
            }
#endif
            pl = plN;
            rcN = rcB; endN = endB;
        }
        if (ferr) break;
        if (nbSeq && nValid < nbSeq) { ferr = E_corruption; break; }
        if (nbSeq && tailErr) { ferr = tailErr; break; }
        {   // last literals (:1308-1317)
            uint32_t const rest = litSize - litPos;
            if (pos > room || rest > room - pos) { ferr = E_dstSize_tooSmall; break; }
            uint8_t* const d = fdst + bo + pos;
            if (litRle) { uint8_t const v = litBase[0]; for (uint32_t i = lane; i < rest; i += 64) d[i] = v; }
            else coop_copy(d, litBase + litPos, rest, lane);
            pos += rest;
        }
        produced += pos;
    }
    if (lane == 0) { fs->regenSize = produced; fs->errorCode = ferr; }
}

// ---------------------------------------------------------------- stage 3b: execute, eight waves per frame
// The sequences of a block must take effect in order, but most of the work of a batch of 64 does not depend on the
// batches just before it; k_execute above leaves its one wave idle through every wait.  Here the waves of a workgroup
// take the batches of a block in turn (wave w: batches w, w + 8, ...) and assemble them in ONE ring of the block's
// recent output, in LDS, shared by the workgroup.  Per batch:
//   early (nothing waited for)  records (requested one batch earlier), plan; literal bytes and the match sources that
//                       lie far back come from memory into the ring; matches whose source lies in completed batches are
//                       copied ring to ring.  For what has to wait, the moves are worked out byte by byte and kept in
//                       registers: the bytes of the matches whose source is still being assembled are dealt out four to
//                       a lane; each match that depends on the batch itself (or repeats a pattern) gets a round of one
//                       byte per lane.
//   late (in turn)      once every earlier batch is complete -- they complete in order -- the dealt bytes move in ONE
//                       round trip to LDS, then the dependent matches one round each, in sequence order.
//   after               the batch is marked complete; its bytes go from the ring to memory with whole-wave stores, and
//                       are announced as flushed when the wave comes round to its next batch (the stores have landed
//                       by then: they precede the record loads the wave has to wait for anyway).
// What the waves wait for one another on is the late step alone.  Control words in LDS (polled): {batches complete,
// output position complete, ring floor, error} and {batches flushed, output position flushed}.
// A batch with too much output for its share of the ring goes from memory to memory as in k_execute, after everything
// before it has been flushed; it raises the ring floor past itself (its bytes are not in the ring).  A batch with a
// late source outside the ring's reach does its late part the slow way: match after match in sequence order, from
// memory where it must.
constexpr int kWf = 8;
constexpr uint32_t kWfMaxFrames = 1024;   // batches of more frames than this are decoded by k_execute alone
constexpr uint32_t kWfRing = 32768, kWfMask = kWfRing - 1;
constexpr uint32_t kWfHist = 16384;       // a match source at most this far before its batch is read from the ring
constexpr uint32_t kWfStage = 3072;       // most output bytes of a batch assembled in the ring
constexpr uint32_t kWfLitStage = 2048;    // most literal bytes of such a batch
enum { WF_DONE = 0, WF_DONEPOS, WF_FLOOR, WF_ERR, WF_FLUSHED, WF_FLUSHEDPOS };
constexpr uint32_t kWfAbort = 0x7FFFFFFFu;
#ifndef ZG_WF_NAP
#define ZG_WF_NAP 1
#endif
constexpr int kWfNap = ZG_WF_NAP;         // s_sleep between polls of the control words: a polling wave leaves the LDS and its SIMD to the others
constexpr uint32_t kWfTabWords = 64;      // head bytes[256] of the list of late bytes
enum { WFC_NONE = 0, WFC_EARLY, WFC_LATE_RING, WFC_DEPENDENT, WFC_LATE_MEMORY };

DEVI void wf_barrier()
{
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Two runs of 0..32 bytes into the ring, at block-relative positions dA and dB: run A from a linear LDS buffer, run B
// from one or (bRing) from the ring at position sB.  A run that crosses the ring's end goes byte by byte.
DEVI void wf_copy_runs(uint8_t* ring, uint32_t dA, const uint8_t* srcA, uint32_t lenA, uint32_t dB, const uint8_t* srcB, bool bRing, uint32_t sB, uint32_t lenB)
{
    uint32_t const a = dA & kWfMask, b = dB & kWfMask, sb = sB & kWfMask;
    bool const wrapA = a + lenA > kWfRing;
    bool const wrapB = (b + lenB > kWfRing) | (bRing & (sb + lenB > kWfRing));
    lds_copy_runs(ring + a, srcA, wrapA ? 0u : lenA, ring + b, bRing ? ring + sb : srcB, wrapB ? 0u : lenB);
    if (wrapA) for (uint32_t j = 0; j < lenA; j++) ring[(dA + j) & kWfMask] = srcA[j];
    if (wrapB) for (uint32_t j = 0; j < lenB; j++) ring[(dB + j) & kWfMask] = bRing ? ring[(sB + j) & kWfMask] : srcB[j];
}

// inclusive prefix maximum over the 64 lanes of a wave (all lanes active)
DEVI uint32_t wave_incl_max(uint32_t v)
{
    v = max(v, dpp_or0<0x111, 0xF>(v)); v = max(v, dpp_or0<0x112, 0xF>(v)); v = max(v, dpp_or0<0x114, 0xF>(v)); v = max(v, dpp_or0<0x118, 0xF>(v));
    v = max(v, dpp_or0<0x142, 0xA>(v));
    v = max(v, dpp_or0<0x143, 0xC>(v));
    return v;
}
// value of the next lower lane (0 in lane 0)
DEVI uint32_t wave_prev0(uint32_t x) { return dpp_or0<0x138, 0xF>(x); }

// The late step of a batch.  First, four single bytes per lane ring to ring, every load before the first store and no
// branch -- one LDS round trip for the whole wave; no source is another lane's destination.  Then the dependent matches
// in sequence order, each one byte per lane (source | destination << 16).  A lane with nothing to move moves its own
// spare byte behind the ring onto itself.  Offsets are worked out by the caller ahead of the wait for the batch's turn.
constexpr int kWfDeps = 6;
struct WfLateStep {
    uint32_t d[4], s[4];          // ring offsets
    uint32_t dep[kWfDeps];
    DEVI void pin()
    {
        // (worked out here and now, not after the wait that follows)
        asm volatile("" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]));
        asm volatile("" : "+v"(dep[0]), "+v"(dep[1]), "+v"(dep[2]), "+v"(dep[3]), "+v"(dep[4]), "+v"(dep[5]));
    }
    DEVI void run_list(uint8_t* ring) const
    {
        uint8_t const v0 = ring[s[0]], v1 = ring[s[1]], v2 = ring[s[2]], v3 = ring[s[3]];
        ring[d[0]] = v0; ring[d[1]] = v1; ring[d[2]] = v2; ring[d[3]] = v3;
    }
    DEVI void run_deps(uint8_t* ring, uint32_t n) const
    {
#pragma unroll
        for (int k = 0; k < kWfDeps; k++)
            if ((uint32_t)k < n) { uint8_t const v = ring[dep[k] & 0xFFFF]; ring[dep[k] >> 16] = v; }     // (LDS operations of a wave execute in order)
    }
};

// Polls LDS control words until `cond`; what follows is not moved above it.  The wait is bounded (far beyond what a
// block can take): should the protocol ever stall, the frame ends with an error instead of a wave that never finishes
// (a verdict already written -- a real error of the frame that made the others stop -- is kept).
#define WF_SPIN(cond)                                                                                   \
    do {                                                                                                \
        uint32_t spins_ = 0;                                                                            \
        while (!(cond)) {                                                                               \
            __builtin_amdgcn_s_sleep(kWfNap);                                                           \
            if (++spins_ > (1u << 22)) { if (!ctl[WF_ERR]) ctl[WF_ERR] = E_generic; ctl[WF_DONE] = kWfAbort; stalled = true; break; } \
        }                                                                                               \
        asm volatile("" ::: "memory");                                                                  \
    } while (0)

__global__ __launch_bounds__(64 * kWf) __attribute__((amdgpu_waves_per_eu(4))) void k_execute_wf(BatchParams p)
{
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    typedef uint32_t v2u __attribute__((ext_vector_type(2)));
    __shared__ __attribute__((aligned(16))) uint8_t ring[kWfRing + 64 * kWf];        // + a spare byte per lane
    __shared__ __attribute__((aligned(16))) uint8_t litStageAll[kWf][kWfLitStage];
    __shared__ __attribute__((aligned(16))) uint8_t matStageAll[kWf][64 * 32];
    __shared__ __attribute__((aligned(16))) uint32_t tabAll[kWf][kWfTabWords];
    __shared__ __attribute__((aligned(16))) uint32_t ctlMem[8];
    // (typed as LDS on purpose: a volatile access through a generic pointer is a flat memory operation, many times slower)
    typedef __attribute__((address_space(3))) volatile uint32_t LdsWord;
    typedef __attribute__((address_space(3))) volatile v2u LdsPair;
    typedef __attribute__((address_space(3))) volatile v4u LdsQuad;
    LdsWord* const ctl = (LdsWord*)ctlMem;
    uint32_t const tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    uint8_t* const litStage = litStageAll[wave];
    uint8_t* const matStage = matStageAll[wave];
    uint32_t* const tHeads = tabAll[wave];
    // sequence table of a batch that goes from memory to memory (it uses neither stage)
    uint32_t* const sStart = (uint32_t*)matStage;
    uint32_t* const sLL = sStart + 65;
    uint32_t* const sOff = sLL + 64;
    uint32_t* const sLit = sOff + 64;
    uint32_t* const sSlowEnd = sLit + 64;
    uint32_t const fi = blockIdx.x;
    DevFrame const f = p.frames[fi];
    DevFrameStatus* const fs = p.status + fi;
    if (fs->errorCode) return;
    if (!frame_uses_wf(p, f)) return;
    uint8_t* const fdst = p.dst + f.dstOff;
    uint64_t const cap = f.dstCapacity;
    uint32_t const dictN = f.useDict ? p.dictContentSize : 0;
    const uint8_t* const dictEnd = p.dictContent + dictN;
    uint32_t ferr = 0;
    uint64_t produced = 0;

    for (uint32_t b = f.firstBlock; b < f.firstBlock + f.nBlocks && !ferr; b++) {
        const DevBlock* const bp = p.blocks + b;
        const DevBlockState* const sp_ = p.state + b;
        uint32_t const blockType = uni((uint32_t)bp->blockType), origSize = uni(bp->origSize);
        uint64_t const blkSrcOff = uni(bp->srcOff);
        uint32_t const hostLitErr = uni((uint32_t)bp->hostLitErr), hostSeqErr = uni((uint32_t)bp->hostSeqErr);
        uint32_t const litType = uni((uint32_t)bp->litType), nbSeq = uni(bp->nbSeq);
        uint64_t const bo = produced;
        uint64_t const room = cap - bo;
        if (blockType == BT_STOP) { ferr = hostLitErr; break; }
        if (blockType == BT_RAW || blockType == BT_RLE) {
            if (origSize > room) { ferr = E_dstSize_tooSmall; break; }
            if (blockType == BT_RAW) { const uint8_t* const sp = p.src + blkSrcOff; for (uint64_t i = tid; i < origSize; i += 64 * kWf) fdst[bo + i] = sp[i]; }
            else { uint8_t const v = p.src[blkSrcOff]; for (uint64_t i = tid; i < origSize; i += 64 * kWf) fdst[bo + i] = v; }
            produced += origSize;
            wf_barrier();       // in memory before a later block reads them
            continue;
        }
        if (hostLitErr) { ferr = hostLitErr; break; }
        bool const hufLits = litType >= SET_COMPRESSED;
        uint32_t const litErr = uni(sp_->litErr), seqErr = uni(sp_->seqErr), tailErr = uni(sp_->tailErr);
        if (hufLits && litErr) { ferr = litErr; break; }
        if (hostSeqErr) { ferr = hostSeqErr; break; }
        if (nbSeq && seqErr) { ferr = seqErr; break; }

        const uint8_t* const litBase = hufLits ? p.litBuf + uni(bp->litBufOff) : p.src + blkSrcOff + uni((uint32_t)bp->litHdrSize);
        bool const litRle = (litType == SET_RLE);
        uint32_t const litSize = uni(bp->litSize);
        uint32_t const rleFill = (litRle && litSize) ? 0x01010101u * uni((uint32_t)litBase[0]) : 0u;
        uint32_t const nValid = nbSeq ? uni(sp_->seqValid) : 0;
        const uint4* const recs = (const uint4*)p.seqBuf + uni(bp->seqBufOff);
        uint32_t const rin0 = uni(sp_->repIn[0]), rin1 = uni(sp_->repIn[1]), rin2 = uni(sp_->repIn[2]);
        uint32_t const nBatches = (nValid + 63) >> 6;
        uint8_t* const bdst = fdst + bo;                  // the block's output

        wf_barrier();           // every wave is through with the block before (its control words, its flushes)
        if (tid < 8) ctlMem[tid] = 0;
        wf_barrier();

        // records of this wave's next batch, requested a batch ahead (past the last batch: nothing)
        v4u rN = {0, 0, 0, 0};
        v2u eN = {0, 0};
        auto load_recs = [&](uint32_t bt2) {
            if (bt2 < nBatches) {
                uint32_t const s = bt2 << 6;
                uint32_t const n2 = nValid - s < 64 ? nValid - s : 64;
                rN = __builtin_nontemporal_load((const v4u*)(recs + s + (lane < n2 ? lane : n2)));
                eN = __builtin_nontemporal_load((const v2u*)((const uint32_t*)(recs + s + n2) + 1));
            }
        };
        load_recs(wave);
        bool stalled = false;
        bool flushPending = false;          // this wave's last batch is on its way to memory, not yet announced
        uint32_t fpBt = 0, fpEnd = 0;
        // announces the flush of the wave's previous batch: its stores have landed, and every earlier batch is announced
        auto announce = [&]() {
            if (flushPending) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                WF_SPIN(uni(ctl[WF_FLUSHED]) >= fpBt);
                if (lane == 0) { v2u const w = {fpBt + 1, fpEnd}; *(LdsPair*)(ctlMem + WF_FLUSHED) = w; }
                flushPending = false;
            }
        };
        for (uint32_t bt = wave; bt < nBatches && !stalled; bt += kWf) {
            uint32_t const s0 = bt << 6;
            uint32_t const n = nValid - s0 < 64 ? nValid - s0 : 64;
#ifdef ZG_STAMPS
            uint64_t const y0 = __builtin_readcyclecounter();
            uint64_t y1 = y0, y2 = y0, y3 = y0, y4 = y0, y5 = y0;
            uint32_t nLate = 0, nSlow = 0, nFarLate = 0, wentSlow = 0;
            uint64_t z1 = y0, z2 = y0, z3 = y0, z4 = y0, z5 = y0, z6 = y0;
#endif
            announce();
#ifdef ZG_STAMPS
            z1 = __builtin_readcyclecounter();
#endif
            // ---- records and plan (see k_execute: the same per-sequence facts and checks)
            v4u const r4 = rN;
            v2u const e2v = eN;
            bool const have = lane < n;
            uint32_t const endY = uni(e2v.x), endZ = uni(e2v.y);
            uint32_t const below = wave_next(r4.y);
            uint32_t const nstart = lane == 63 ? endY : (have ? below : r4.y);
            uint32_t const bstart = r4.y;
            uint32_t const lstart = r4.z;
            uint32_t const ll = have ? r4.w : 0;
            uint32_t const offR = rep_subst(r4.x, rin0, rin1, rin2);
            uint32_t const off = have ? offR : 1;
            bool const e1 = nstart > room;
            bool const e2 = (uint64_t)lstart + ll > litSize;
            bool const e3 = (uint64_t)off > bo + bstart + ll + dictN;
            uint32_t const code = !have ? 0u : e1 ? (uint32_t)E_dstSize_tooSmall : (e2 | e3) ? (uint32_t)E_corruption : 0u;
            unsigned long long const badm = __ballot(code != 0);
            uint32_t const perr = badm ? uni((uint32_t)__shfl(code, __builtin_ctzll(badm))) : 0u;
            uint32_t const ml = nstart - bstart - ll;
            uint32_t const base0 = lane_value(bstart, 0);
            uint32_t const start = bstart - base0;
            uint32_t const total = lane_value(nstart, 63) - base0;
            uint32_t const lit0 = lane_value(lstart, 0);
            uint32_t const sumLL = endZ - lit0;
            if (perr) {
                // the first bad sequence in order ends the block: wait for this batch's turn, then tell everybody
                WF_SPIN(ctl[WF_DONE] >= bt);
                if (ctl[WF_DONE] == bt) { ctl[WF_ERR] = perr; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ctl[WF_DONE] = kWfAbort; }
                break;
            }
            int64_t const ms = (int64_t)(start + ll) - (int64_t)off;           // batch-relative source start
            int64_t const sp = (int64_t)base0 + ms;                             // block-relative
            int64_t const absS = (int64_t)bo + sp;                              // frame-relative (negative: dictionary)
            bool const plain = have & (off >= ml) & (ms + (int64_t)ml <= 0) & ((absS >= 0) | (absS + (int64_t)ml <= 0));
            int64_t const fetchEnd = absS + (ml > 16 ? 32 : 16);
            bool const fetchable = plain & (absS >= 0 ? fetchEnd <= (int64_t)cap : fetchEnd <= 0);
            bool const staged = (total <= kWfStage) & (sumLL <= kWfLitStage);
            const uint8_t* const matSrc = absS >= 0 ? fdst + absS : dictEnd + absS;

            if (!staged) {
                // ---- memory to memory, once everything before the batch is in memory
                WF_SPIN(ctl[WF_DONE] == kWfAbort || (ctl[WF_DONE] >= bt && ctl[WF_FLUSHED] >= bt));
                if (ctl[WF_DONE] != bt) break;         // an earlier batch failed
#ifdef ZG_STAMPS
                y1 = __builtin_readcyclecounter();
#endif
                uint32_t const fastLenD = plain ? ml : 0, slowLen = (have && !plain) ? ml : 0;
                uint32_t const slowEnd = wave_incl_scan(slowLen);
                uint32_t const slowTotal = lane_value(slowEnd, 63);
                lds_sync();
                sStart[lane] = start; sLL[lane] = ll; sOff[lane] = off; sLit[lane] = lstart; sSlowEnd[lane] = slowEnd;
                if (lane == 63) sStart[64] = total;
                lds_sync();
                exec_batch_direct(sStart, sLL, sOff, sLit, sSlowEnd, litBase, fdst, dictEnd, (int64_t)(bo + base0), litRle, bdst + base0 + start,
                                  litRle ? litBase : litBase + lstart, matSrc, ll, fastLenD, slowTotal, bdst + base0, lane);
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                if (lane == 0) {
                    ctl[WF_FLOOR] = base0 + total;          // nothing of this batch is in the ring
                    v2u const w = {bt + 1, base0 + total};
                    *(LdsPair*)(ctlMem + WF_FLUSHED) = w;
                    *(LdsPair*)(ctlMem + WF_DONE) = w;
                }
#ifdef ZG_STAMPS
                if (p.dbg && lane == 0) {
                    unsigned long long* const g = (unsigned long long*)(p.dbg + 32 * (size_t)p.nBlocks + 32 * fi);
                    atomicAdd(g + 10, 1ull); atomicAdd(g + 11, y1 - y0); atomicAdd(g + 12, __builtin_readcyclecounter() - y1);
                }
#endif
                load_recs(bt + kWf);
                continue;
            }

#ifdef ZG_STAMPS
            z2 = __builtin_readcyclecounter();
#endif
            // ---- early part.  The ring must not be overwritten under a batch still being assembled (it may look kWfHist
            // back) nor before its bytes have been flushed: both hold once this batch ends within ring - history of the
            // completed and within a ring of the flushed output.
            WF_SPIN(uni(ctl[WF_DONE]) == kWfAbort || (base0 + total <= uni(ctl[WF_DONEPOS]) + (kWfRing - kWfHist) && base0 + total <= uni(ctl[WF_FLUSHEDPOS]) + kWfRing));
            v4u const c0 = *(LdsQuad*)ctlMem;
            v2u const f0 = *(LdsPair*)(ctlMem + WF_FLUSHED);
            if (uni(c0.x) == kWfAbort) break;
            uint32_t const donePos0 = uni(c0.y), floor0 = uni(c0.z), flushedPos0 = uni(f0.y);
            bool const nearS = plain & (sp >= (int64_t)floor0) & (sp + (int64_t)kWfHist >= (int64_t)base0);   // source in the ring's reach
            bool const nearNow = nearS & (sp + (int64_t)ml <= (int64_t)donePos0);           // ... and complete already
            bool const farS = fetchable & !nearS;
            bool const farNow = farS & (sp + (int64_t)ml <= (int64_t)flushedPos0);          // in memory already
            bool const slowS = have & !nearS & !farS & (ml != 0);
            uint32_t const cls = (ml == 0 || !have) ? (uint32_t)WFC_NONE : (nearNow | farNow) ? (uint32_t)WFC_EARLY : nearS ? (uint32_t)WFC_LATE_RING
                                 : farS ? (uint32_t)WFC_LATE_MEMORY : (uint32_t)WFC_DEPENDENT;
            uint32_t const litFull = sumLL >> 4, litTail = sumLL & 15;
            const uint8_t* const safe = (const uint8_t*)recs;
            const uint8_t* const litG = litBase + lit0;
            bool const farFetch = farNow & (ml <= 32);          // (longer ones are copied piece by piece further down)
            uint4 litV, mv0, mv1;
            __builtin_memcpy(&litV, (!litRle && lane < litFull) ? litG + 16 * lane : safe, 16);
            uint8_t const litT = *((!litRle && lane < litTail) ? litG + 16 * litFull + lane : safe);
            __builtin_memcpy(&mv0, farFetch ? matSrc : safe, 16);
            __builtin_memcpy(&mv1, (farFetch && ml > 16) ? matSrc + 16 : safe, 16);

            bool const lateRing = nearS & !nearNow & (ml <= 32);          // (longer ones go with the long dependent matches)
#ifdef ZG_STAMPS
            z3 = __builtin_readcyclecounter();
#endif
            // ---- while those are in flight: the dependent matches.  The late step copies them one after the other in sequence
            // order, one byte per lane; which byte from where is worked out here (byte j of a match copies pattern byte
            // j mod offset; the pattern is the `offset` bytes before the match).  A late match of more than 32 bytes goes the
            // same way.  Up to kWfDeps matches of up to 64 bytes, their patterns within the ring's reach; a batch with more
            // does its late part the slow way.
            uint32_t const dLit = base0 + start, dMat = base0 + start + ll;          // block-relative destinations
            bool const dep = slowS | (nearS & !nearNow & (ml > 32));
            bool const depBad = dep & ((ml > 64) | (sp < (int64_t)floor0) | (sp + (int64_t)kWfHist < (int64_t)base0));
            unsigned long long const deps = __ballot(dep);
            uint32_t const nDeps = (uint32_t)__builtin_popcountll(deps);
            bool slowWay = __any((cls == WFC_LATE_MEMORY) | depBad) != 0;
#ifdef ZG_STAMPS
            uint32_t why = (__any(cls == WFC_LATE_MEMORY) ? 2u : 0u) | (__any(depBad) ? 4u : 0u);
#endif
            unsigned long long depsRest = deps;         // the dependent matches past the first kWfDeps: copied by a loop
            // the lowest ring position this lane's part of the late step reads (should the floor rise above it in the meantime,
            // those bytes are in memory, not in the ring)
            uint32_t const lowSrc = (lateRing | dep) ? (uint32_t)sp : 0xFFFFFFFFu;
            uint32_t const spare = kWfRing + 64 * wave + lane;
            WfLateStep late;
            for (int k = 0; k < 4; k++) late.d[k] = late.s[k] = spare;
            for (int k = 0; k < kWfDeps; k++) late.dep[k] = spare | (spare << 16);
            if (!slowWay) {
#pragma unroll
                for (int k = 0; k < kWfDeps; k++) {
                    if (depsRest) {
                        int const w = __builtin_ctzll(depsRest);
                        depsRest &= depsRest - 1;
                        uint32_t const d0 = lane_value(dMat, w), o = lane_value(off, w), m = lane_value(ml, w);
                        uint32_t r = lane;
                        if (o < 64) {       // lane mod offset: the quotient by a reciprocal, off by one at most
                            uint32_t const q = (uint32_t)((float)lane * __builtin_amdgcn_rcpf((float)o));
                            r = lane - q * o;
                            r = (int32_t)r < 0 ? r + o : r;
                            r = r >= o ? r - o : r;
                        }
                        if (lane < m) late.dep[k] = ((d0 - o + r) & kWfMask) | (((d0 + lane) & kWfMask) << 16);
                    }
                }
            }

#ifdef ZG_STAMPS
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            z4 = __builtin_readcyclecounter();
#endif
            // ---- the bytes of the matches whose source is still being assembled, dealt out four to a lane (lane l: bytes
            // 4l .. 4l+3 of their concatenation).  Each such match marks its first byte with its lane; a prefix maximum
            // carries the marks forward.
            bool lateRun = false;
            {
                uint32_t const lrLen = lateRing ? ml : 0;
                uint32_t const lrEnd = wave_incl_scan(lrLen);
                // (the list holds 256 bytes; the matches past that are copied as runs after it)
                unsigned long long const fits = __ballot(lrEnd <= 256);
                uint32_t const lrTotal = fits ? lane_value(lrEnd, 63 - __builtin_clzll(fits)) : 0u;
                lateRun = lateRing & (lrEnd > 256);
                if (lrTotal && !slowWay) {
                    tHeads[lane] = 0;
                    if (lrLen && lrEnd <= 256) ((uint8_t*)tHeads)[lrEnd - lrLen] = (uint8_t)(lane + 1);
                    lds_sync();
                    uint32_t const h = tHeads[lane];
                    uint32_t o0 = h & 0xFF, o1 = (h >> 8) & 0xFF, o2 = (h >> 16) & 0xFF, o3 = h >> 24;
                    o1 = o1 ? o1 : o0; o2 = o2 ? o2 : o1; o3 = o3 ? o3 : o2;
                    uint32_t const carry = wave_prev0(wave_incl_max(o3));
                    uint32_t const own[4] = {o0 ? o0 : carry, o1 ? o1 : carry, o2 ? o2 : carry, o3 ? o3 : carry};
                    uint32_t const packed = ((base0 + start + ll) & kWfMask) | ((lrEnd - lrLen) << 16);      // where the match goes in the ring | its first byte in the list
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        int const from = (int)(own[k] ? own[k] - 1 : 0);
                        uint32_t const a = (uint32_t)__shfl((int)packed, from), o = (uint32_t)__shfl((int)off, from);
                        uint32_t const e = 4 * lane + k;
                        uint32_t const dd = ((a & 0xFFFF) + (e - (a >> 16))) & kWfMask;
                        bool const act = e < lrTotal;
                        late.d[k] = act ? dd : spare;
                        late.s[k] = act ? (dd - o) & kWfMask : spare;
                    }
                }
            }

#ifdef ZG_STAMPS
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            z5 = __builtin_readcyclecounter();
#endif
            lds_sync();           // this wave's previous reads of its stages are over
            if (litRle) *(uint4*)(litStage + 16 * lane) = make_uint4(rleFill, rleFill, rleFill, rleFill);
            else {
                if (lane < litFull) *(uint4*)(litStage + 16 * lane) = litV;
                if (litFull > 64) {     // (rare: the second kilobyte)
                    uint4 litW;
                    __builtin_memcpy(&litW, lane + 64 < litFull ? litG + 16 * (lane + 64) : safe, 16);
                    if (lane + 64 < litFull) *(uint4*)(litStage + 16 * (lane + 64)) = litW;
                }
                if (lane < litTail) litStage[16 * litFull + lane] = litT;
            }
            if (farFetch) { *(uint4*)(matStage + 32 * lane) = mv0; if (ml > 16) *(uint4*)(matStage + 32 * lane + 16) = mv1; }
            lds_sync();
            wf_copy_runs(ring, dLit, litStage + (litRle ? 0u : lstart - lit0), ll <= 32 ? ll : 0u,
                         dMat, matStage + 32 * lane, nearNow, (uint32_t)(dMat - off), ((nearNow | farNow) & (ml <= 32)) ? ml : 0u);
#ifdef ZG_STAMPS
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            z6 = __builtin_readcyclecounter();
#endif
            // ---- runs of more than 32 bytes, one after the other, by the whole wave in 16-byte pieces (the last piece ends
            // where the run ends); a run that crosses the end of the ring goes byte by byte
            {
                bool const longMat = (nearNow | farNow) & (ml > 32);
                unsigned long long lm = __ballot((ll > 32) | longMat);
                while (lm) {
                    int const w = __builtin_ctzll(lm);
                    lm &= lm - 1;
                    uint32_t const wl = lane_value(ll, w), wm = lane_value(ml, w), wo = lane_value(off, w);
                    uint32_t const wd = lane_value(dLit, w), kind = lane_value((uint32_t)longMat | ((uint32_t)nearNow << 1), w);
                    if (wl > 32) {
                        uint32_t const so = lane_value(lstart - lit0, w);
                        if (litRle) for (uint32_t c = lane; c < wl; c += 64) ring[(wd + c) & kWfMask] = (uint8_t)rleFill;
                        else if ((wd & kWfMask) + wl > kWfRing) for (uint32_t c = lane; c < wl; c += 64) ring[(wd + c) & kWfMask] = litStage[so + c];
                        else for (uint32_t c = 16 * lane; c < wl; c += 1024) {
                            uint32_t const cc = c + 16 <= wl ? c : wl - 16;
                            uint4 x;
                            __builtin_memcpy(&x, litStage + so + cc, 16);
                            __builtin_memcpy(ring + ((wd + cc) & kWfMask), &x, 16);
                        }
                    }
                    if (kind & 1) {
                        uint32_t const md = wd + wl;         // the match's place
                        if (kind & 2) {     // from completed batches, ring to ring (source and destination do not overlap)
                            uint32_t const ms0 = md - wo;
                            if ((md & kWfMask) + wm > kWfRing || (ms0 & kWfMask) + wm > kWfRing) for (uint32_t c = lane; c < wm; c += 64) ring[(md + c) & kWfMask] = ring[(ms0 + c) & kWfMask];
                            else for (uint32_t c = 16 * lane; c < wm; c += 1024) {
                                uint32_t const cc = c + 16 <= wm ? c : wm - 16;
                                uint4 x;
                                __builtin_memcpy(&x, ring + ((ms0 + cc) & kWfMask), 16);
                                __builtin_memcpy(ring + ((md + cc) & kWfMask), &x, 16);
                            }
                        } else {            // from memory
                            const uint8_t* const g = (const uint8_t*)(uintptr_t)(((uint64_t)lane_value((uint32_t)((uintptr_t)matSrc >> 32), w) << 32) | lane_value((uint32_t)(uintptr_t)matSrc, w));
                            if ((md & kWfMask) + wm > kWfRing) for (uint32_t c = lane; c < wm; c += 64) ring[(md + c) & kWfMask] = g[c];
                            else for (uint32_t c = 16 * lane; c < wm; c += 1024) {
                                uint32_t const cc = c + 16 <= wm ? c : wm - 16;
                                uint4 x;
                                __builtin_memcpy(&x, g + cc, 16);
                                __builtin_memcpy(ring + ((md + cc) & kWfMask), &x, 16);
                            }
                        }
                    }
                }
            }
#ifdef ZG_STAMPS
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            y2 = __builtin_readcyclecounter();
#endif
            // ---- late part: every earlier batch is complete
            late.pin();
            bool const anyLateRun = __any(lateRun) != 0;
            v4u c1;
            WF_SPIN(uni((c1 = *(LdsQuad*)ctlMem).x) >= bt);
#ifdef ZG_STAMPS
            y3 = __builtin_readcyclecounter();
#endif
            if (stalled || uni(c1.x) != bt) break;
            uint32_t const floor1 = uni(c1.z);
            if (!slowWay && (floor1 == floor0 || !__any(lowSrc < floor1))) {
                late.run_list(ring);
                if (anyLateRun) wf_copy_runs(ring, 0u, ring, 0u, dMat, ring, true, (uint32_t)(dMat - off), lateRun ? ml : 0u);
                late.run_deps(ring, nDeps);
                while (depsRest) {
                    int const w = __builtin_ctzll(depsRest);
                    depsRest &= depsRest - 1;
                    uint32_t const d0 = lane_value(dMat, w), o = lane_value(off, w), m = lane_value(ml, w);
                    if (lane < m) ring[(d0 + lane) & kWfMask] = ring[(d0 - o + (o < m ? lane % o : lane)) & kWfMask];
                }
            } else {
#ifdef ZG_STAMPS
                wentSlow = 1;
                if (!slowWay) why |= 32u;
                if (p.dbg && lane == 0) {
                    unsigned long long* const g = (unsigned long long*)(p.dbg + 32 * (size_t)p.nBlocks + 32 * fi);
                    for (int k = 0; k < 6; k++) if (why >> k & 1) atomicAdd(g + 16 + k, 1ull);
                }
#endif
                bool const lateNear = nearS & !nearNow & (sp >= (int64_t)floor1);
                bool const lateFar = (farS & !farNow) | (nearS & !nearNow & !lateNear);        // (a source below a floor raised meanwhile is in memory, not in the ring)
                bool slowNeedsMem = false;
                if (slowS) slowNeedsMem = sp < (int64_t)floor1 || sp + (int64_t)kWfHist < (int64_t)base0;
                if (__any(lateFar | slowNeedsMem)) {
                    WF_SPIN(ctl[WF_FLUSHED] >= bt);
                    // (rare: byte by byte; the conditions of `fetchable` do not all hold for a source that was near)
                    if (lateFar) for (uint32_t j = 0; j < ml; j++) ring[(dMat + j) & kWfMask] = matSrc[j];
                }
                wf_copy_runs(ring, 0u, ring, 0u, dMat, ring, true, (uint32_t)(dMat - off), (lateNear & (ml <= 32)) ? ml : 0u);
                // the dependent matches (and the long ones of the matches just copied) one after the other in sequence order: all that lies before a match is in place when
                // its turn comes (LDS operations of a wave execute in order).  Byte j copies pattern byte j mod offset; the
                // pattern is the `offset` bytes before the match.
                unsigned long long sm = __ballot(slowS | (lateNear & (ml > 32)));
                while (sm) {
                    int const w = __builtin_ctzll(sm);
                    sm &= sm - 1;
                    uint32_t const d0 = lane_value(dMat, w), o = lane_value(off, w), m = lane_value(ml, w);
                    bool const mem = lane_value((uint32_t)slowNeedsMem, w) != 0;
                    int64_t const src0 = (int64_t)d0 - (int64_t)o;              // block-relative start of the pattern
                    for (uint32_t j = lane; j < m; j += 64) {
                        int64_t const q = src0 + (int64_t)(o < m ? j % o : j);
                        uint8_t v;
                        if (mem && (q < (int64_t)floor1 || q + (int64_t)kWfHist < (int64_t)base0)) {
                            int64_t const a = (int64_t)bo + q;
                            v = a >= 0 ? fdst[a] : dictEnd[a];
                        } else v = ring[(uint32_t)q & kWfMask];
                        ring[(d0 + j) & kWfMask] = v;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) { v2u const w = {bt + 1, base0 + total}; *(LdsPair*)(ctlMem + WF_DONE) = w; }
#ifdef ZG_STAMPS
            y4 = __builtin_readcyclecounter();
            uint64_t const prevDone = ((uint64_t)ctl[7] << 32) | ctl[6];
            if (lane == 0) { ctl[6] = (uint32_t)y4; ctl[7] = (uint32_t)(y4 >> 32); }
            if (p.dbg && lane == 0 && bt != 0 && prevDone) {
                unsigned long long* const g = (unsigned long long*)(p.dbg + 32 * (size_t)p.nBlocks + 32 * fi);
                atomicAdd(g + 14, y4 - prevDone); atomicAdd(g + 15, y3 - prevDone);
            }
            nLate = (uint32_t)__builtin_popcountll(__ballot(lateRing)); nSlow = (uint32_t)__builtin_popcountll(__ballot(slowS)); nFarLate = (uint32_t)__builtin_popcountll(__ballot(cls == WFC_LATE_MEMORY));
#endif
            // ---- flush: 16 bytes per lane at ring-aligned addresses, the ragged ends byte by byte; announced later
            {
                uint32_t const a0 = base0, a1 = base0 + total;
                uint32_t const u0 = (a0 + 15) & ~15u, u1 = a1 & ~15u;
                if (u0 <= u1) {
                    for (uint32_t q = u0 + 16 * lane; q < u1; q += 1024) {
                        uint4 const x = *(const uint4*)(ring + (q & kWfMask));
                        __builtin_memcpy(bdst + q, &x, 16);
                    }
                    if (a0 + lane < u0) bdst[a0 + lane] = ring[(a0 + lane) & kWfMask];
                    if (u1 + lane < a1) bdst[u1 + lane] = ring[(u1 + lane) & kWfMask];
                } else if (a0 + lane < a1) bdst[a0 + lane] = ring[(a0 + lane) & kWfMask];
            }
            asm volatile("" ::: "memory");
            flushPending = true; fpBt = bt; fpEnd = base0 + total;
            load_recs(bt + kWf);
#ifdef ZG_STAMPS
            y5 = __builtin_readcyclecounter();
            if (p.dbg && lane == 0) {
                unsigned long long* const g = (unsigned long long*)(p.dbg + 32 * (size_t)p.nBlocks + 32 * fi);
                atomicAdd(g + 0, y2 - y0); atomicAdd(g + 1, y3 - y2); atomicAdd(g + 2, y4 - y3); atomicAdd(g + 3, y5 - y4);
                atomicAdd(g + 5, (unsigned long long)wentSlow);
                atomicAdd(g + 6, 1ull); atomicAdd(g + 7, (unsigned long long)nLate); atomicAdd(g + 8, (unsigned long long)nSlow); atomicAdd(g + 9, (unsigned long long)nFarLate);
                atomicAdd(g + 22, z1 - y0); atomicAdd(g + 23, z2 - z1); atomicAdd(g + 24, z3 - z2); atomicAdd(g + 25, z4 - z3); atomicAdd(g + 26, z5 - z4); atomicAdd(g + 27, z6 - z5); atomicAdd(g + 28, y2 - z6);
            }
#endif
        }
        announce();
        wf_barrier();
        {
            uint32_t const err = ctl[WF_ERR];
            if (err) { ferr = err; break; }
        }
        if (nbSeq && nValid < nbSeq) { ferr = E_corruption; break; }
        if (nbSeq && tailErr) { ferr = tailErr; break; }
        {   // last literals (:1308-1317), by the whole workgroup
            uint2 endRec = make_uint2(0, 0);
            if (nValid) { const uint32_t* const er = (const uint32_t*)(recs + nValid); endRec = make_uint2(uni(er[1]), uni(er[2])); }
            uint64_t const pos = endRec.x;
            uint32_t const litPos = endRec.y;
            uint32_t const rest = litSize - litPos;
            if (pos > room || rest > room - pos) { ferr = E_dstSize_tooSmall; break; }
            uint8_t* const d = bdst + pos;
            if (litRle) { uint8_t const v = litBase[0]; for (uint32_t i = tid; i < rest; i += 64 * kWf) d[i] = v; }
            else for (uint32_t i = tid; i < rest; i += 64 * kWf) d[i] = litBase[litPos + i];       // (byte-wise on purpose: the blocks this kernel takes end in few literals, and the wide copy costs registers)
            produced += pos + rest;
        }
    }
    if (tid == 0) { fs->regenSize = produced; fs->errorCode = ferr; }
}

// ---------------------------------------------------------------- stage 4: XXH64 + verdict
constexpr uint64_t P1 = 0x9E3779B185EBCA87ULL, P2 = 0xC2B2AE3D27D4EB4FULL, P3 = 0x165667B19E3779F9ULL,
                   P4 = 0x85EBCA77C2B2AE63ULL, P5 = 0x27D4EB2F165667C5ULL;
DEVI uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
DEVI uint64_t xxh_round(uint64_t acc, uint64_t in) { acc += in * P2; acc = rotl64(acc, 31); return acc * P1; }
DEVI uint64_t xxh_merge(uint64_t acc, uint64_t v) { v = xxh_round(0, v); acc ^= v; return acc * P1 + P4; }

// XXH64 of [p, p+len) with seed 0, computed by a group of 8 consecutive lanes (half a DPP row); the result is valid
// on the group's first lane (xxhash.pas:409-488).  A round is acc = rotl(acc + in*P2, 31) * P1: two 64-bit multiplies,
// which are most of its issue time, and only the second depends on the round before.  So the group's lanes 0..3 carry
// the four accumulators and lanes 7..4 prepare in*P2 of the following stripe for them; both do it with the same
// multiply instruction (their own factor in K), and the product crosses over through a half-row mirror, which pairs
// lane w with lane 7-w.  After step t a carrying lane holds the accumulator after t+1 rounds, a preparing lane in*P2 of
// stripe t+1.
#ifndef ZG_XXH_DEPTH
#define ZG_XXH_DEPTH 16
#endif
constexpr int kXxhDepth = ZG_XXH_DEPTH;        // stripes requested ahead of the one being folded in
// y + mirror(y): the two halves of a 64-bit addition with the cross-lane read folded into the additions themselves
// (the two wait states are what a DPP read of a register the instruction before wrote needs)
DEVI uint64_t add_half_mirror64(uint64_t y)
{
    uint32_t const yl = (uint32_t)y, yh = (uint32_t)(y >> 32);
    uint32_t lo, hi;
    asm volatile("s_nop 1\n\t"
                 "v_add_co_u32_dpp %0, vcc, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "v_addc_co_u32_dpp %1, vcc, %3, %3, vcc row_half_mirror row_mask:0xf bank_mask:0xf"
                 : "=&v"(lo), "=&v"(hi) : "v"(yl), "v"(yh) : "vcc");
    return ((uint64_t)hi << 32) | lo;
}
// (the choice between the two roles as a bit select: written as a conditional it becomes a branch around the addition)
#define XXH_STEP(IN) { uint64_t const a_ = add_half_mirror64(y);                                                           \
                       uint32_t const al_ = (uint32_t)a_, ah_ = (uint32_t)(a_ >> 32);                                      \
                       uint32_t const xl_ = __builtin_amdgcn_alignbit(al_, ah_, 1), xh_ = __builtin_amdgcn_alignbit(ah_, al_, 1); \
                       uint32_t const il_ = (uint32_t)(IN), ih_ = (uint32_t)((IN) >> 32);                                  \
                       y = (((uint64_t)((ih_ & role) | (xh_ & ~role)) << 32) | ((il_ & role) | (xl_ & ~role))) * K; }
#define XXH_LOAD(BUF, FIRST) _Pragma("unroll") for (int k = 0; k < kXxhDepth; k++) BUF[k] = ld_u64(q + 32 * ((FIRST) + k));
#define XXH_FOLD(BUF) _Pragma("unroll") for (int k = 0; k < kXxhDepth; k++) XXH_STEP(BUF[k]);

DEVI uint64_t xxh64_group8(const uint8_t* p, uint64_t len, uint32_t gl, uint32_t lead)
{
    uint64_t h;
    uint64_t const stripes = len / 32;
    bool const prep = gl >= 4;
    uint32_t const w = prep ? 7 - gl : gl;                  // which of the stripe's four words this lane works on
    uint64_t const K = prep ? P2 : P1;
    uint32_t const role = prep ? ~0u : 0u;
    const uint8_t* const q = p + 8 * w;
    uint64_t y = w == 0 ? P1 + P2 : w == 1 ? P2 : w == 2 ? 0 : 0 - P1;
    if (stripes) {
        if (prep) y = ld_u64(q) * P2;
        uint64_t s = 0;                                     // steps done; step t takes the word of stripe t+1
        // The steps of a lane depend on one another, so what a wave can hide of the memory latency is what it requests
        // ahead: kXxhDepth stripes are in flight while the kXxhDepth before them are folded in.
        if (s + kXxhDepth < stripes) {
            uint64_t bufA[kXxhDepth], bufB[kXxhDepth];          // (two buffers taking turns, so that no step pays for a register copy)
            XXH_LOAD(bufA, s + 1)
            for (;;) {
                if (!(s + 2 * kXxhDepth < stripes)) { XXH_FOLD(bufA) s += kXxhDepth; break; }
                XXH_LOAD(bufB, s + kXxhDepth + 1)
                XXH_FOLD(bufA)
                s += kXxhDepth;
                if (!(s + 2 * kXxhDepth < stripes)) { XXH_FOLD(bufB) s += kXxhDepth; break; }
                XXH_LOAD(bufA, s + kXxhDepth + 1)
                XXH_FOLD(bufB)
                s += kXxhDepth;
            }
        }
        // 1..kXxhDepth steps are left; a word past the last stripe is read from the last stripe instead (its product is not used)
        {
            uint64_t t[kXxhDepth];
            uint64_t const last = stripes - 1;
#pragma unroll
            for (int k = 0; k < kXxhDepth; k++) { uint64_t const i = s + 1 + k; t[k] = ld_u64(q + 32 * (i < last ? i : last)); }
#pragma unroll
            for (int k = 0; k < kXxhDepth; k++) if (s + k < stripes) XXH_STEP(t[k]);
        }
    }
    uint64_t const v1 = __shfl(y, lead), v2 = __shfl(y, lead + 1), v3 = __shfl(y, lead + 2), v4 = __shfl(y, lead + 3);
    if (len >= 32) {
        h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        h = xxh_merge(h, v1); h = xxh_merge(h, v2); h = xxh_merge(h, v3); h = xxh_merge(h, v4);
    } else h = P5;
    h += len;
    const uint8_t* t = p + stripes * 32;
    const uint8_t* const end = p + len;
    if (gl == 0) {
        while (t + 8 <= end) { h ^= xxh_round(0, ld_u64(t)); h = rotl64(h, 27) * P1 + P4; t += 8; }
        if (t + 4 <= end) { uint32_t w4; __builtin_memcpy(&w4, t, 4); h ^= (uint64_t)w4 * P1; h = rotl64(h, 23) * P2 + P3; t += 4; }
        while (t < end) { h ^= (uint64_t)(*t) * P5; h = rotl64(h, 11) * P1; t++; }
        h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    }
    return h;
}
#undef XXH_STEP
#undef XXH_LOAD
#undef XXH_FOLD

__global__ __launch_bounds__(64) void k_checksum(BatchParams p)
{
    uint32_t const lane = threadIdx.x;
    uint32_t const gl = lane & 7, lead = lane & ~7u;
    uint32_t const fi = blockIdx.x * 8 + (lane >> 3);
    if (fi >= p.nFrames) return;
    DevFrame const f = p.frames[fi];
    DevFrameStatus* const fs = p.status + fi;
    uint32_t err = fs->errorCode;
    uint64_t const regen = fs->regenSize;
    // frame verdict (zstd_decompressf.pas:810-828): content size, then checksum
    if (!err && f.contentSize != ~0ull && regen != f.contentSize) err = E_corruption;
    bool const doSum = !err && f.checksumFlag != 0;
    uint64_t h = 0;
    if (doSum && f.checksumFlag == 1) h = xxh64_group8(p.dst + f.dstOff, regen, gl, lead);
    if (gl == 0) {
        if (doSum) {
            if (f.checksumFlag == 2) err = E_checksum_wrong;          // trailer missing from the source
            else { fs->xxh64Low32 = (uint32_t)h; if ((uint32_t)h != f.checksum) err = E_checksum_wrong; }
        }
        fs->errorCode = err;
    }
}

__global__ __launch_bounds__(64) void k_xxh64(const uint8_t* base, const uint64_t* off, const uint64_t* len, uint32_t n, uint64_t* out)
{
    uint32_t const lane = threadIdx.x;
    uint32_t const gl = lane & 7, lead = lane & ~7u;
    uint32_t const i = blockIdx.x * 8 + (lane >> 3);
    if (i >= n) return;
    uint64_t const h = xxh64_group8(base + off[i], len[i], gl, lead);
    if (gl == 0) out[i] = h;
}

// ---------------------------------------------------------------- dictionary digest (single lane)
// Zero fill of a 16-byte aligned area whose size is a multiple of 16 (a kernel instead of a memset node: see launch_zero's callers)
__global__ __launch_bounds__(256) void k_zero(uint4* p, size_t units)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < units; i += (size_t)gridDim.x * 256) p[i] = make_uint4(0, 0, 0, 0);
}

__global__ __launch_bounds__(64) void k_digest_dict(const uint8_t* dict, uint32_t dictSize, DevDict* out)
{
    __shared__ __attribute__((aligned(16))) uint8_t scratch[2048];
    __shared__ uint8_t weights[256];
    __shared__ uint16_t pos[256];
    __shared__ uint32_t rank[16];
    __shared__ int16_t norm[64];
    __shared__ uint16_t symNext[64];
    if (threadIdx.x != 0) return;
    out->status = E_dictionary_corrupted;
    out->hasEntropy = 1;
    out->hufX1 = 0; out->entropyValid = 3;
    if (dictSize <= 8) return;
    uint32_t o = 8;
    {   // Huffman table for literals; the reference builds the double-symbol form here (:1288), which decodes the same bytes
        uint32_t nbSymbols = 0, tableLog = 0;
        uint32_t const h = dev_huf_read_table_desc(dict + o, dictSize - o, weights, pos, rank, scratch, nbSymbols, tableLog);
        if (h == kFail) return;
        for (uint32_t n = 0; n < nbSymbols; n++) {
            uint32_t const w = weights[n];
            if (!w) continue;
            uint32_t const len = (1u << w) >> 1;
            uint16_t const cell = (uint16_t)(n | ((tableLog + 1 - w) << 8));
            for (uint32_t u = 0; u < len; u++) out->huf[pos[n] + u] = cell;
        }
        out->hufLog = tableLog;
        o += h;
    }
    // OF, ML, LL in that order (:1294-1340)
    for (int k = 0; k < 3; k++) {
        int const t = k == 0 ? T_OF : k == 1 ? T_ML : T_LL;
        uint32_t* const cells = t == T_LL ? out->ll : t == T_OF ? out->of : out->ml;
        uint32_t maxSV = tbl_max(t), tableLog = 0;
        if (o > dictSize) return;
        uint32_t const h = dev_readNCount(norm, maxSV, tableLog, dict + o, dictSize - o);
        if (h == kFail || tableLog > tbl_maxLog(t)) return;
        dev_build_seq_table(t, cells, norm, symNext, maxSV, tableLog);
        if (t == T_LL) out->llLog = tableLog; else if (t == T_OF) out->ofLog = tableLog; else out->mlLog = tableLog;
        o += h;
    }
    if (o + 12 > dictSize) return;
    uint32_t const contentSize = dictSize - (o + 12);
    for (int i = 0; i < 3; i++) {
        uint32_t const rep = rd32z(dict, dictSize, o); o += 4;
        if (rep == 0 || rep > contentSize) return;
        out->rep[i] = rep;
    }
    out->entropySize = o;
    out->status = 0;
}

// ---------------------------------------------------------------- block-at-a-time decoding (single lane)
// ZSTD_decompressContinue (zstd_decompressf.pas:1137-1207) hands the decoder one block per call; what a block leaves
// behind for the next one is the context's entropy state (zstd_ddict.pas:109-126): the Huffman table and the three
// sequence tables last defined, and the repeat offsets.  The host decodes such a block as the only block of a frame
// whose "dictionary" is that state (`carry`, in the dictionary's cell formats; its content is the output so far); this
// kernel, run behind the block's stage chain, moves the tables the block defined and its final repeat offsets into the
// state.  Frame 0 of the launch, its first block.
__global__ __launch_bounds__(64) void k_carry_entropy(BatchParams p, DevDict* carry)
{
    __shared__ __attribute__((aligned(16))) uint8_t scratch[2048];
    __shared__ uint8_t weights[256];
    __shared__ uint16_t pos[256];
    __shared__ uint32_t rank[16];
    __shared__ int16_t norm[64];
    __shared__ uint16_t symNext[64];
    if (threadIdx.x != 0) return;
    if (batch_overflow(p) || p.status[0].errorCode) return;     // (the caller gives up the frame on an error: the state no longer matters)
    DevFrame const f = p.frames[0];
    if (f.hostError || f.nBlocks == 0) return;
    uint32_t const b = f.firstBlock;
    DevBlock const blk = p.blocks[b];
    if (blk.blockType != BT_COMPRESSED) return;
    const uint8_t* const payload = p.src + blk.srcOff;
    if (blk.litType == SET_COMPRESSED) {
        // HUF_readDTableX1 / X2 on the block's own description (zstd_decompress_block.pas:333-366); one cell format here
        uint32_t nbSymbols = 0, tableLog = 0;
        uint32_t const h = dev_huf_read_table_desc(payload + blk.litHdrSize, blk.litCSize, weights, pos, rank, scratch, nbSymbols, tableLog);
        if (h != kFail) {
            for (uint32_t n = 0; n < nbSymbols; n++) {
                uint32_t const w = weights[n];
                if (!w) continue;
                uint32_t const len = (1u << w) >> 1;
                uint16_t const cell = (uint16_t)(n | ((tableLog + 1 - w) << 8));
                for (uint32_t u = 0; u < len; u++) carry->huf[pos[n] + u] = cell;
            }
            carry->hufLog = tableLog;
            // which decoder built it decides how a later block that repeats the table judges its streams' last symbols
            carry->hufX1 = (!blk.singleStream && blk.litSize != 0 && huf_select_x2(blk.litSize, blk.litCSize)) ? 0u : 1u;
            carry->entropyValid |= 1u;
        }
    }
    if (blk.nbSeq) {
        uint32_t o = blk.seqHdrOff;
        bool bad = false;
        for (int k = 0; k < 3 && !bad; k++) {       // descriptions in the order LL, OF, ML (:744-781)
            int const t = k == 0 ? T_LL : k == 1 ? T_OF : T_ML;
            uint32_t const type = t == T_LL ? blk.llType : t == T_OF ? blk.ofType : blk.mlType;
            if (type == SET_REPEAT) continue;
            uint32_t* const cells = t == T_LL ? carry->ll : t == T_OF ? carry->of : carry->ml;
            uint32_t lg = 0;
            if (o > blk.cSize) { bad = true; break; }
            uint32_t const h = dev_build_from_desc(t, type, payload + o, blk.cSize - o, cells, norm, symNext, lg);
            if (h == kFail) { bad = true; break; }
            if (t == T_LL) carry->llLog = lg; else if (t == T_OF) carry->ofLog = lg; else carry->mlLog = lg;
            o += h;
        }
        if (!bad) carry->entropyValid |= 2u;
        const DevBlockState* const st = p.state + b;
        uint32_t const in[3] = {st->repIn[0], st->repIn[1], st->repIn[2]};
        uint32_t const n0 = rep_resolve(st->repOut[0], in), n1 = rep_resolve(st->repOut[1], in), n2 = rep_resolve(st->repOut[2], in);
        carry->rep[0] = n0; carry->rep[1] = n1; carry->rep[2] = n2;
    }
}

// XXH64 in pieces (XXH64_reset / XXH64_update / XXH64_digest, xxhash.pas:533-545, :686-757, :759-828), for output that
// arrives block by block: lanes 0..3 carry the four accumulators; after every piece the digest of everything so far is
// left in the state, so the host reads it when the frame's checksum arrives.
__global__ __launch_bounds__(64) void k_xxh64_stream(DevXxhState* s, const uint8_t* data, uint64_t len)
{
    uint32_t const w = threadIdx.x;
    if (w >= 4) return;
    uint64_t acc = s->v[w];
    uint32_t const memsize = s->memsize;
    uint64_t const total = s->total + len;
    uint8_t mem[32];
#pragma unroll
    for (int i = 0; i < 32; i++) mem[i] = s->mem[i];
    uint32_t newMem;
    if (memsize + len < 32) {
        for (uint32_t i = 0; i < (uint32_t)len; i++) mem[memsize + i] = data[i];
        newMem = memsize + (uint32_t)len;
    } else {
        const uint8_t* q = data;
        uint64_t left = len;
        if (memsize) {
            uint32_t const take = 32 - memsize;
            for (uint32_t i = 0; i < take; i++) mem[memsize + i] = q[i];
            uint64_t word = 0;
            for (int j = 0; j < 8; j++) word |= (uint64_t)mem[8 * w + j] << (8 * j);
            acc = xxh_round(acc, word);
            q += take; left -= take;
        }
        uint64_t const stripes = left / 32;
        for (uint64_t i = 0; i < stripes; i++) acc = xxh_round(acc, ld_u64(q + 32 * i + 8 * w));
        q += 32 * stripes; left -= 32 * stripes;
        for (uint32_t i = 0; i < (uint32_t)left; i++) mem[i] = q[i];
        newMem = (uint32_t)left;
    }
    uint64_t const v1 = __shfl(acc, 0), v2 = __shfl(acc, 1), v3 = __shfl(acc, 2), v4 = __shfl(acc, 3);
    s->v[w] = acc;
    if (w == 0) {
        s->total = total; s->memsize = newMem;
        for (uint32_t i = 0; i < newMem; i++) s->mem[i] = mem[i];
        uint64_t h;
        if (total >= 32) {
            h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
            h = xxh_merge(h, v1); h = xxh_merge(h, v2); h = xxh_merge(h, v3); h = xxh_merge(h, v4);
        } else h = v3 + P5;
        h += total;
        uint32_t t = 0;
        while (t + 8 <= newMem) { uint64_t word = 0; for (int j = 0; j < 8; j++) word |= (uint64_t)mem[t + j] << (8 * j); h ^= xxh_round(0, word); h = rotl64(h, 27) * P1 + P4; t += 8; }
        if (t + 4 <= newMem) { uint32_t w4 = 0; for (int j = 0; j < 4; j++) w4 |= (uint32_t)mem[t + j] << (8 * j); h ^= (uint64_t)w4 * P1; h = rotl64(h, 23) * P2 + P3; t += 4; }
        while (t < newMem) { h ^= (uint64_t)mem[t] * P5; h = rotl64(h, 11) * P1; t++; }
        h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
        s->digest = h;
    }
}

}  // namespace

// ---------------------------------------------------------------- launchers
void launch_literals(const BatchParams& p, void* stream)
{
    if (!p.nLitJobs) return;
    if (p.litDictSingleStream && p.dict) hipLaunchKernelGGL(k_literals<true>, dim3((p.nLitJobs + kHufDictJobs - 1) / kHufDictJobs), dim3(64 * kHufDictWaves), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(k_literals<false>, dim3((p.nLitJobs + kHufJobs - 1) / kHufJobs), dim3(64), 0, (hipStream_t)stream, p);
}
void launch_pack_literals(const BatchParams& p, void* stream)
{
    if (!p.nLitJobs) return;
    hipLaunchKernelGGL(k_pack_literals, dim3(p.nLitJobs), dim3(kPackThreads), 0, (hipStream_t)stream, p);
}
void launch_sequences(const BatchParams& p, void* stream)
{
    if (!p.nSeqJobs) return;
    dim3 const grid((p.nSeqJobs + kSeqJobs - 1) / kSeqJobs);
    if (p.seqTablesFromDict && p.dict) hipLaunchKernelGGL(k_sequences<true>, grid, dim3(kSeqLaunchThreadsDict), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(k_sequences<false>, grid, dim3(kSeqThreads), 0, (hipStream_t)stream, p);
}
void launch_sequences_dry(const BatchParams& p, void* stream)
{
    if (!p.nSeqJobs) return;
    uint32_t const grid = p.nSeqJobs < 128 ? p.nSeqJobs : 128;
    hipLaunchKernelGGL(k_sequences_dry, dim3(grid), dim3(64), 0, (hipStream_t)stream, p);
}
void launch_layout(const BatchParams& p, void* stream)
{
    if (!p.nFrames) return;
    hipLaunchKernelGGL(k_layout, dim3((p.nFrames + 63) / 64), dim3(64), 0, (hipStream_t)stream, p);
}
// pIn.execMode: 1 or 2 sends every frame to one of the two execute kernels (zstdgpu_test_setExecuteKernel: tests,
// measurements); 0 leaves the choice to the shape of the batch and of each frame
void launch_execute(const BatchParams& pIn, void* stream)
{
    if (!pIn.nFrames) return;
    BatchParams p = pIn;
    // Eight waves per frame pay off while the chip has room for them: two such workgroups fit a compute unit, and past
    // four rounds of them one wave per frame (all frames resident at once) is the faster kernel.
    uint32_t const forced = pIn.execMode <= 2 ? pIn.execMode : 0u;
    p.execMode = forced ? forced : (p.nFrames > kWfMaxFrames ? 1u : 0u);
    if (p.execMode != 1) hipLaunchKernelGGL(k_execute_wf, dim3(p.nFrames), dim3(64 * kWf), 0, (hipStream_t)stream, p);
    if (p.execMode != 2) {
        if (p.dict) hipLaunchKernelGGL(k_execute<true>, dim3(p.nFrames), dim3(64), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL(k_execute<false>, dim3(p.nFrames), dim3(64), 0, (hipStream_t)stream, p);
    }
}
void launch_checksum(const BatchParams& p, void* stream)
{
    if (!p.nFrames) return;
    hipLaunchKernelGGL(k_checksum, dim3((p.nFrames + 7) / 8), dim3(64), 0, (hipStream_t)stream, p);
}
void launch_digest_dict(const uint8_t* devDict, uint32_t dictSize, DevDict* out, void* stream)
{
    hipLaunchKernelGGL(k_digest_dict, dim3(1), dim3(64), 0, (hipStream_t)stream, devDict, dictSize, out);
}
void launch_basic_tables(uint32_t* out, void* stream)
{
    hipLaunchKernelGGL(k_basic_tables, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
}
void launch_zero(void* p, size_t bytes, void* stream)
{
    if (!bytes) return;
    size_t const units = bytes / 16;
    uint32_t const grid = (uint32_t)((units + 255) / 256 < 1024 ? (units + 255) / 256 : 1024);
    hipLaunchKernelGGL(k_zero, dim3(grid ? grid : 1), dim3(256), 0, (hipStream_t)stream, (uint4*)p, units);
}
void launch_xxh64(const uint8_t* base, const uint64_t* off, const uint64_t* len, uint32_t n, uint64_t* out, void* stream)
{
    if (!n) return;
    hipLaunchKernelGGL(k_xxh64, dim3((n + 7) / 8), dim3(64), 0, (hipStream_t)stream, base, off, len, n, out);
}
void launch_carry_entropy(const BatchParams& p, DevDict* carry, void* stream)
{
    hipLaunchKernelGGL(k_carry_entropy, dim3(1), dim3(64), 0, (hipStream_t)stream, p, carry);
}
void launch_xxh64_stream(DevXxhState* state, const uint8_t* data, uint64_t len, void* stream)
{
    hipLaunchKernelGGL(k_xxh64_stream, dim3(1), dim3(64), 0, (hipStream_t)stream, state, data, len);
}

}  // namespace zg
