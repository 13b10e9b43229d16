// Block and frame descriptors built on the device, from a source buffer that is already in device memory.
//
// The same walk as the host builder (zstdgpu_host.cpp), one lane per frame:
//   frame header      ZSTD_frameHeaderSize_internal / ZSTD_getFrameHeader_advanced (decompress/zstd_decompressf.pas:246-395)
//   block headers     ZSTD_getcBlockSize (decompress/zstd_decompress_block.pas:190-207), the loop of
//                     ZSTD_findFrameSizeInfo (decompress/zstd_decompressf.pas:566-639)
//   section headers   the header switch of ZSTD_decodeLiteralsBlock (decompress/zstd_decompress_block.pas:243-274,
//                     :333-400) and the count / modes bytes of ZSTD_decodeSeqHeaders (:708-742)
//   table links       which block defines the Huffman / LL / OF / ML table in use (the entropy state that
//                     ZSTD_DCtx carries from block to block, decompress/zstd_ddict.pas:109-126)
// Frames are independent, so the walk is parallel over frames; inside a frame it is serial (a block header tells
// where the next one is).  Three launches: count the blocks of every frame, fill the block descriptors, then give
// every block its place in the scratch buffers (prefix sums over the blocks) and list the literal / sequence jobs.
#include <hip/hip_runtime.h>
#include "zstdgpu_internal.h"

namespace zg {
namespace {

#define DEVI __device__ __forceinline__

constexpr uint32_t kMagic = 0xFD2FB528u;            // zstd.pas:5
constexpr uint64_t kContentSizeUnknown = 0xFFFFFFFFFFFFFFFFull;
constexpr uint32_t kWindowLogMax = 31;
constexpr uint32_t kLongNbSeq = 0x7F00;

DEVI uint32_t g16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
DEVI uint32_t g24(const uint8_t* p) { return g16(p) | ((uint32_t)p[2] << 16); }
DEVI uint32_t g32(const uint8_t* p) { return g24(p) | ((uint32_t)p[3] << 24); }
DEVI uint64_t g64(const uint8_t* p) { return (uint64_t)g32(p) | ((uint64_t)g32(p + 4) << 32); }

// One frame: header, then every block header.  emit(i, payloadOffset, cSize, origSize, type, last, error) is called for
// every block slot in order.  Fills the frame descriptor except firstBlock; `slots` = block slots the frame takes in
// the block array (they stay even when the frame ends up with an error and nBlocks = 0, as in the host builder).
template <typename Emit>
DEVI void walk_frame(const uint8_t* base, uint64_t srcBytes, FrameIn const in, uint32_t batchDictID, uint32_t opts,
                     DevFrame& f, uint32_t& slots, Emit emit)
{
    f.dstOff = in.dstOff; f.dstCapacity = in.dstCapacity; f.contentSize = kContentSizeUnknown;
    f.firstBlock = 0; f.nBlocks = 0; f.checksumFlag = 0; f.checksum = 0; f.hostError = 0; f.useDict = opts & kOptDict; f.longWindow = 0; f.pad = 0;
    uint32_t const pre = (opts & kOptMagicless) ? 1u : 5u;      // bytes up to and including the descriptor byte
    slots = 0;
    if (in.srcOff > srcBytes || in.srcSize > srcBytes - in.srcOff) { f.hostError = E_srcSize_wrong; return; }
    const uint8_t* const src = base + in.srcOff;
    uint64_t remaining = in.srcSize;
    if (remaining < pre + 1 + 3) { f.hostError = E_srcSize_wrong; return; }
    uint32_t const fhd = src[pre - 1];
    uint32_t const dictIDSizeCode = fhd & 3, singleSegment = (fhd >> 5) & 1, fcsID = fhd >> 6;
    uint32_t const fhsize = pre + (singleSegment ? 0u : 1u) + (dictIDSizeCode == 3 ? 4u : dictIDSizeCode) +
                            (fcsID == 0 ? 0u : fcsID == 1 ? 2u : fcsID == 2 ? 4u : 8u) + ((singleSegment && !fcsID) ? 1u : 0u);
    if (remaining < (uint64_t)fhsize + 3) { f.hostError = E_srcSize_wrong; return; }
    if (pre == 5 && g32(src) != kMagic) { f.hostError = E_prefix_unknown; return; }
    if (fhd & 0x08) { f.hostError = E_frameParameter_unsupported; return; }
    uint32_t pos = pre;
    uint64_t windowSize = 0;
    if (!singleSegment) {
        uint32_t const wl = src[pos++];
        if ((wl >> 3) + 10 > kWindowLogMax) { f.hostError = E_frameParameter_windowTooLarge; return; }
        windowSize = 1ull << ((wl >> 3) + 10);
        windowSize += (windowSize >> 3) * (wl & 7);
    }
    uint32_t dictID = 0;
    if (dictIDSizeCode == 1) { dictID = src[pos]; pos += 1; }
    else if (dictIDSizeCode == 2) { dictID = g16(src + pos); pos += 2; }
    else if (dictIDSizeCode == 3) { dictID = g32(src + pos); pos += 4; }
    uint64_t contentSize = kContentSizeUnknown;
    if (fcsID == 0) { if (singleSegment) contentSize = src[pos]; }
    else if (fcsID == 1) contentSize = g16(src + pos) + 256;
    else if (fcsID == 2) contentSize = g32(src + pos);
    else contentSize = g64(src + pos);
    f.contentSize = contentSize;
    if (singleSegment) windowSize = contentSize;          // zstd_decompressf.pas:375
    f.longWindow = windowSize > (1ull << 24) ? 1u : 0u;
    uint64_t ip = fhsize;
    remaining -= fhsize;
    bool stopped = false;
    for (;;) {
        bool done;
        if (remaining < 3) { emit(slots, (uint64_t)0, 0u, 0u, (uint32_t)BT_STOP, 1u, (uint32_t)E_srcSize_wrong); stopped = true; done = true; }
        else {
            uint32_t const hdr = g24(src + ip);
            uint32_t const type = (hdr >> 1) & 3, size = hdr >> 3;
            uint32_t const cSize = type == BT_RLE ? 1 : size;
            if (type == 3) { emit(slots, (uint64_t)0, 0u, 0u, (uint32_t)BT_STOP, 1u, (uint32_t)E_corruption); stopped = true; done = true; }
            else if (cSize > remaining - 3) { emit(slots, (uint64_t)0, 0u, 0u, (uint32_t)BT_STOP, 1u, (uint32_t)E_srcSize_wrong); stopped = true; done = true; }
            else {
                emit(slots, in.srcOff + ip + 3, cSize, size, type, hdr & 1, 0u);
                ip += 3 + cSize; remaining -= 3 + cSize;
                done = (hdr & 1) != 0;
            }
        }
        slots++;
        if (done) break;
    }
    f.nBlocks = slots;
    if (!stopped && ((fhd >> 2) & 1)) {
        if (remaining < 4) f.checksumFlag = 2;   // trailer missing: checksum_wrong once everything else passed
        else { f.checksumFlag = (opts & kOptIgnoreChecksum) ? 0u : 1u; f.checksum = g32(src + ip); ip += 4; }
    }
    // one entry is exactly one frame: bytes after it are an error
    if (ip != in.srcSize && f.checksumFlag != 2 && !stopped) f.hostError = E_srcSize_wrong;
    if (!f.hostError && dictID && dictID != batchDictID) f.hostError = E_dictionary_wrong;     // ZSTD_decodeFrameHeader :541-563
    if (!f.hostError && f.dstCapacity > 0x7FFFFFFEull) f.dstCapacity = 0x7FFFFFFEull;           // frames regenerate < 2 GiB
    if (f.hostError) { f.nBlocks = 0; f.longWindow = 0; }
}

// Literals-section header and the count / modes bytes of the sequences section of one compressed block.
DEVI void parse_sections(const uint8_t* payload, DevBlock& d)
{
    uint32_t const cSize = d.cSize;
    if (cSize >= kBlockSizeMax) { d.hostLitErr = E_srcSize_wrong; return; }       // zstd_decompress_block.pas:1513
    if (cSize < 3) { d.hostLitErr = E_corruption; return; }                        // MIN_CBLOCK_SIZE
    uint32_t const type = payload[0] & 3, lhl = (payload[0] >> 2) & 3;
    d.litType = (uint8_t)type;
    uint32_t section = 0;
    if (type >= SET_COMPRESSED) {
        if (type == SET_REPEAT && d.hufSrc == SRC_NONE) { d.hostLitErr = E_dictionary_corrupted; return; }   // :233-234
        if (cSize < 5) { d.hostLitErr = E_corruption; return; }
        uint32_t const lhc = g32(payload);
        uint32_t lhSize, litSize, litCSize;
        if (lhl < 2) { d.singleStream = lhl ? 0 : 1; lhSize = 3; litSize = (lhc >> 4) & 0x3FF; litCSize = (lhc >> 14) & 0x3FF; }
        else if (lhl == 2) { lhSize = 4; litSize = (lhc >> 4) & 0x3FFF; litCSize = lhc >> 18; }
        else { lhSize = 5; litSize = (lhc >> 4) & 0x3FFFF; litCSize = (lhc >> 22) + ((uint32_t)payload[4] << 10); }
        if (litSize > kBlockSizeMax) { d.hostLitErr = E_corruption; return; }
        if (litCSize + lhSize > cSize) { d.hostLitErr = E_corruption; return; }
        d.litHdrSize = (uint8_t)lhSize; d.litSize = litSize; d.litCSize = litCSize;
        section = lhSize + litCSize;
    } else {
        uint32_t lhSize, litSize;
        if (lhl == 1) { lhSize = 2; litSize = g16(payload) >> 4; }
        else if (lhl == 3) {
            lhSize = 3; litSize = g24(payload) >> 4;
            if (type == SET_RLE && cSize < 4) { d.hostLitErr = E_corruption; return; }
        } else { lhSize = 1; litSize = payload[0] >> 3; }
        d.litHdrSize = (uint8_t)lhSize; d.litSize = litSize;
        if (type == SET_BASIC) {
            if (litSize + lhSize > cSize) { d.hostLitErr = E_corruption; return; }
            d.litCSize = litSize; section = lhSize + litSize;
        } else {
            if (litSize > kBlockSizeMax) { d.hostLitErr = E_corruption; return; }
            d.litCSize = 1; section = lhSize + 1;
        }
    }
    // sequences header
    uint32_t const rem = cSize - section;
    uint32_t ip = section;
    if (rem < 1) { d.hostSeqErr = E_srcSize_wrong; return; }
    uint32_t nbSeq = payload[ip++];
    if (!nbSeq) { if (rem != 1) d.hostSeqErr = E_srcSize_wrong; d.nbSeq = 0; return; }
    if (nbSeq > 0x7F) {
        if (nbSeq == 0xFF) { if (ip + 2 > cSize) { d.hostSeqErr = E_srcSize_wrong; return; } nbSeq = g16(payload + ip) + kLongNbSeq; ip += 2; }
        else { if (ip >= cSize) { d.hostSeqErr = E_srcSize_wrong; return; } nbSeq = ((nbSeq - 0x80) << 8) + payload[ip++]; }
    }
    d.nbSeq = nbSeq;
    if (ip + 1 > cSize) { d.hostSeqErr = E_srcSize_wrong; d.nbSeq = 0; return; }
    uint32_t const modes = payload[ip++];
    d.llType = (uint8_t)(modes >> 6); d.ofType = (uint8_t)((modes >> 4) & 3); d.mlType = (uint8_t)((modes >> 2) & 3);
    d.seqHdrOff = ip;
}

// ---- launch 1: frame descriptors (all but firstBlock) and the block slots of every frame
__global__ __launch_bounds__(64) void k_scan_frames(const uint8_t* src, uint64_t srcBytes, const FrameIn* in, uint32_t nFrames,
                                                    uint32_t batchDictID, uint32_t opts, DevFrame* frames, uint32_t* slots)
{
    uint32_t const fi = blockIdx.x * 64 + threadIdx.x;
    if (fi >= nFrames) return;
    DevFrame f;
    uint32_t n;
    walk_frame(src, srcBytes, in[fi], batchDictID, opts, f, n, [](uint32_t, uint64_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t) {});
    frames[fi] = f;
    slots[fi] = n;
}

// Exclusive prefix sum of v over the workgroup's kScanThreads threads (two packed 32-bit counters may ride in one
// 64-bit value as long as neither half overflows inside a tile); `total` = sum over the workgroup.
constexpr int kScanThreads = 1024;
DEVI uint64_t wg_excl_scan(uint64_t v, uint64_t* waveTot, uint64_t& total)
{
    uint32_t const t = threadIdx.x, lane = t & 63, wave = t >> 6;
    uint64_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint64_t const o = ((uint64_t)(uint32_t)__shfl_up((int)(uint32_t)(incl >> 32), d) << 32) | (uint32_t)__shfl_up((int)(uint32_t)incl, d);
        if (lane >= (uint32_t)d) incl += o;
    }
    __syncthreads();                 // the previous use of waveTot is over
    if (lane == 63) waveTot[wave] = incl;
    __syncthreads();
    uint64_t before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < kScanThreads / 64; w++) { uint64_t const x = waveTot[w]; all += x; if ((uint32_t)w < wave) before += x; }
    total = all;
    return before + incl - v;
}

// ---- exclusive prefix sum of the slots into frames[].firstBlock; total -> totals[0].  One workgroup, tiles of
// kScanThreads frames.
__global__ __launch_bounds__(kScanThreads) void k_first_blocks(const uint32_t* slots, const FrameIn* in, uint32_t nFrames, DevFrame* frames, uint64_t* totals,
                                                               uint32_t maxBlocks)
{
    __shared__ uint64_t waveTot[kScanThreads / 64];
    uint32_t const t = threadIdx.x;
    uint64_t carry = 0, bytes = 0;
    for (uint32_t base = 0; base < nFrames; base += kScanThreads) {
        uint32_t const i = base + t;
        bool const in_ = i < nFrames;
        uint64_t const v = in_ ? slots[i] : 0;
        if (in_ && !frames[i].hostError) bytes += in[i].srcSize;
        uint64_t tileTotal;
        uint64_t const ex = wg_excl_scan(v, waveTot, tileTotal);
        if (in_) frames[i].firstBlock = (uint32_t)(carry + ex);
        carry += tileTotal;
    }
    if (bytes) atomicAdd((unsigned long long*)&totals[6], (unsigned long long)bytes);      // source bytes of the frames that will be decoded
    if (t == 0) { totals[0] = carry; if (maxBlocks && carry > maxBlocks) totals[7] = 1; }
}

// ---- launch 2: the block descriptors of every frame.  The scratch offsets hold the block's needs for now
// (litBufOff: literal buffer bytes, litTmpOff: split-decode scratch bytes, seqBufOff: sequence records);
// flags: bit 0 literal job, bit 1 sequence job.  The block array was cleared beforehand: the slots of frames that end up
// with an error stay zero.
__global__ __launch_bounds__(64) void k_fill_blocks(const uint8_t* src, uint64_t srcBytes, const FrameIn* in, uint32_t nFrames,
                                                    uint32_t batchDictID, uint32_t opts, uint32_t dictEntropy, const DevFrame* frames,
                                                    DevBlock* blocks, uint8_t* flags, const uint64_t* totals, uint32_t maxBlocks)
{
    uint32_t const fi = blockIdx.x * 64 + threadIdx.x;
    if (fi >= nFrames) return;
    if (maxBlocks && totals[7]) return;     // more blocks than the fixed block array holds: nothing is filled
    DevFrame const fr = frames[fi];
    if (fr.hostError) return;
    uint32_t const first = fr.firstBlock;
    // entropy state carried from block to block inside the frame: which block defines each table
    // (dictEntropy: bit 0 the dictionary brings a Huffman table, bit 1 it brings the three sequence tables; a digested
    // dictionary brings both, the state carried from block to block by zstdgpu_decompressContinue what the blocks so far set)
    int32_t hufDef = (dictEntropy & 1u) ? SRC_DICT : SRC_NONE;
    int32_t llDef = (dictEntropy & 2u) ? SRC_DICT : SRC_NONE, ofDef = llDef, mlDef = llDef;
    bool dead = false;      // an error found here ends the frame: later blocks are never reached
    DevFrame f;
    uint32_t n;
    walk_frame(src, srcBytes, in[fi], batchDictID, opts, f, n,
               [&](uint32_t i, uint64_t off, uint32_t cSize, uint32_t origSize, uint32_t type, uint32_t last, uint32_t err) {
        uint32_t const b = first + i;
        DevBlock d = {};
        uint8_t fl = 0;
        d.srcOff = off; d.cSize = cSize; d.origSize = origSize; d.frameIdx = fi;
        d.blockType = (uint8_t)type; d.lastBlock = (uint8_t)last;
        d.hufSrc = d.llSrc = d.ofSrc = d.mlSrc = SRC_NONE;
        if (type == BT_STOP) { d.hostLitErr = (uint16_t)(err ? err : (uint32_t)E_corruption); dead = true; }
        else if (dead) { d.blockType = BT_STOP; d.hostLitErr = E_generic; }
        else if (type == BT_COMPRESSED) {
            d.hufSrc = hufDef;
            parse_sections(src + off, d);
            if (d.hostLitErr) dead = true;
            else {
                if (d.litType == SET_COMPRESSED) d.hufSrc = hufDef = (int32_t)b;
                if (d.litType >= SET_COMPRESSED) {
                    d.litBufOff = (d.litSize + 15) & ~15ull;
                    if (!d.singleStream) d.litTmpOff = litTmpBytes(d.litSize);
                    if (!d.singleStream || d.hufSrc != SRC_DICT) atomicAdd((unsigned long long*)&totals[9], 1ull);    // not a single stream on the dictionary's table
                    fl |= 1;
                }
                if (d.hostSeqErr) dead = true;
                else if (d.nbSeq) {
                    if (d.llType != SET_REPEAT) llDef = (int32_t)b;
                    if (d.ofType != SET_REPEAT) ofDef = (int32_t)b;
                    if (d.mlType != SET_REPEAT) mlDef = (int32_t)b;
                    d.llSrc = llDef; d.ofSrc = ofDef; d.mlSrc = mlDef;
                    if (llDef != SRC_DICT || ofDef != SRC_DICT || mlDef != SRC_DICT) atomicAdd((unsigned long long*)&totals[8], 1ull);    // a job with a table of its own
                    d.seqBufOff = (uint64_t)d.nbSeq + 1 + 64;   // + the terminal record + 64 spare records
                    fl |= 2;
                }
            }
        }
        blocks[b] = d;
        flags[b] = fl;
    });
}

// ---- launch 3: every block's place in the scratch buffers and the job lists, in block order.  One workgroup, tiles
// of kScanThreads blocks; inside a tile the literal-buffer and scratch bytes share one 64-bit scan (each < 2^32 per
// tile), the sequence records and the two job counts another.
// totals: [1] literal buffer bytes, [2] split-decode scratch bytes, [3] sequence records, [4] literal jobs, [5] sequence jobs.
__global__ __launch_bounds__(kScanThreads) void k_block_offsets(DevBlock* blocks, const uint8_t* flags, uint32_t nBlocks,
                                                                uint32_t* litJobs, uint32_t* seqJobs, uint64_t* totals, ScratchCaps caps)
{
    __shared__ uint64_t waveTot[kScanThreads / 64];
    uint32_t const t = threadIdx.x;
    if (caps.maxBlocks) {       // fixed scratch: the block count is on the device only
        if (totals[7]) return;
        nBlocks = (uint32_t)totals[0];
    }
    uint64_t cLit = 0, cTmp = 0, cSeq = 0, cLitJobs = 0, cSeqJobs = 0;
    for (uint32_t base = 0; base < nBlocks; base += kScanThreads) {
        uint32_t const b = base + t;
        uint32_t const fl = b < nBlocks ? flags[b] : 0;
        uint64_t lit = 0, tmp = 0, seq = 0;
        if (fl & 1) { lit = blocks[b].litBufOff; tmp = blocks[b].litTmpOff; }
        if (fl & 2) seq = blocks[b].seqBufOff;
        uint64_t totA, totB;
        uint64_t const exA = wg_excl_scan(lit | (tmp << 32), waveTot, totA);
        uint64_t const exB = wg_excl_scan(seq | ((uint64_t)(fl & 1) << 32) | ((uint64_t)((fl >> 1) & 1) << 48), waveTot, totB);
        if (fl & 1) {
            blocks[b].litBufOff = cLit + (uint32_t)exA;
            blocks[b].litTmpOff = tmp ? cTmp + (exA >> 32) : 0;
            litJobs[cLitJobs + ((exB >> 32) & 0xFFFF)] = b;
        }
        if (fl & 2) {
            blocks[b].seqBufOff = cSeq + (uint32_t)exB;
            seqJobs[cSeqJobs + (exB >> 48)] = b;
        }
        cLit += (uint32_t)totA; cTmp += totA >> 32;
        cSeq += (uint32_t)totB; cLitJobs += (totB >> 32) & 0xFFFF; cSeqJobs += totB >> 48;
    }
    if (t == 0) {
        totals[1] = cLit; totals[2] = cTmp; totals[3] = cSeq; totals[4] = cLitJobs; totals[5] = cSeqJobs;
        if (caps.maxBlocks && (cLit + 64 > caps.litBytes || cTmp + 64 > caps.tmpBytes || cSeq + 1 > caps.seqRecords)) totals[7] = 1;
    }
}

}  // namespace

void launch_scan_frames(const uint8_t* src, uint64_t srcBytes, const FrameIn* in, uint32_t nFrames, uint32_t batchDictID, uint32_t opts,
                        DevFrame* frames, uint32_t* slots, uint64_t* totals, void* stream, ScratchCaps caps)
{
    if (nFrames) hipLaunchKernelGGL(k_scan_frames, dim3((nFrames + 63) / 64), dim3(64), 0, (hipStream_t)stream, src, srcBytes, in, nFrames, batchDictID, opts, frames, slots);
    launch_zero(totals, 16 * sizeof(uint64_t), stream);      // (a kernel: a memset node this small is not reliable in a replayed graph)
    hipLaunchKernelGGL(k_first_blocks, dim3(1), dim3(kScanThreads), 0, (hipStream_t)stream, slots, in, nFrames, frames, totals, caps.maxBlocks);
}

void launch_fill_blocks(const uint8_t* src, uint64_t srcBytes, const FrameIn* in, uint32_t nFrames, uint32_t batchDictID, uint32_t opts,
                        uint32_t dictEntropy, const DevFrame* frames, DevBlock* blocks, uint8_t* flags, uint32_t nBlocks,
                        uint32_t* litJobs, uint32_t* seqJobs, uint64_t* totals, void* stream, ScratchCaps caps)
{
    if (nFrames) hipLaunchKernelGGL(k_fill_blocks, dim3((nFrames + 63) / 64), dim3(64), 0, (hipStream_t)stream, src, srcBytes, in, nFrames, batchDictID, opts, dictEntropy, frames, blocks, flags, totals, caps.maxBlocks);
    hipLaunchKernelGGL(k_block_offsets, dim3(1), dim3(kScanThreads), 0, (hipStream_t)stream, blocks, flags, nBlocks, litJobs, seqJobs, totals, caps);
}

}  // namespace zg
