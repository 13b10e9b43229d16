// Host side of the C ABI declared in include/zstdgpu.h: frame/block header walk (the part of
// decompress/zstd_decompressf.pas that stays on the host), section-header parse, table linking,
// dictionary upload, batch management, and the one-shot drop-ins.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/zstdgpu.h"
#include "zstdgpu_internal.h"
#include "zstdgpu_objects.h"

using namespace zg;

namespace {

inline size_t mkerr(uint32_t code) { return (size_t)0 - (size_t)code; }
inline bool iserr(size_t v) { return v > (size_t)0 - (size_t)E_maxCode; }

inline uint32_t rd16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
inline uint32_t rd24(const uint8_t* p) { return rd16(p) | ((uint32_t)p[2] << 16); }
inline uint32_t rd32(const uint8_t* p) { return rd24(p) | ((uint32_t)p[3] << 24); }
inline uint64_t rd64(const uint8_t* p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }

constexpr uint32_t kMagic = 0xFD2FB528u;           // zstd.pas:5
constexpr uint32_t kMagicDict = 0xEC30A437u;       // zstd.pas:6
constexpr uint32_t kMagicSkipStart = 0x184D2A50u;  // zstd.pas:7-8
constexpr uint32_t kMagicSkipMask = 0xFFFFFFF0u;
constexpr uint32_t kWindowLogMax = 31;
constexpr uint32_t kLongNbSeq = 0x7F00;

struct FrameHeader {
    uint64_t contentSize = ZSTDGPU_CONTENTSIZE_UNKNOWN, windowSize = 0;
    uint32_t dictID = 0, checksumFlag = 0, headerSize = 0;
};

// ZSTD_frameHeaderSize_internal (zstd_decompressf.pas:246-279); srcSize >= 5 required
// `magicless`: ZSTD_f_zstd1_magicless, the frame starts at its descriptor byte (ZSTD_startingInputLength :97-107)
size_t frame_header_size(const uint8_t* src, size_t srcSize, bool magicless = false)
{
    static const size_t fcs[4] = {0, 2, 4, 8}, did[4] = {0, 1, 2, 4};
    size_t const pre = magicless ? 1 : 5;
    if (srcSize < pre) return mkerr(E_srcSize_wrong);
    uint8_t const fhd = src[pre - 1];
    uint32_t const singleSegment = (fhd >> 5) & 1, fcsId = fhd >> 6;
    return pre + !singleSegment + did[fhd & 3] + fcs[fcsId] + (singleSegment && !fcsId);
}

// ZSTD_getFrameHeader_advanced (zstd_decompressf.pas:281-395): 0 ok, >0 bytes wanted, or error
size_t get_frame_header(FrameHeader& h, const uint8_t* src, size_t srcSize, bool magicless = false)
{
    size_t const pre = magicless ? 1 : 5;
    if (srcSize < pre) return pre;
    if (!magicless && rd32(src) != kMagic) return mkerr(E_prefix_unknown);
    size_t const fhsize = frame_header_size(src, srcSize, magicless);
    if (srcSize < fhsize) return fhsize;
    h.headerSize = (uint32_t)fhsize;
    uint8_t const fhd = src[pre - 1];
    size_t pos = pre;
    uint32_t const dictIDSizeCode = fhd & 3, singleSegment = (fhd >> 5) & 1, fcsID = fhd >> 6;
    h.checksumFlag = (fhd >> 2) & 1;
    if (fhd & 0x08) return mkerr(E_frameParameter_unsupported);
    if (!singleSegment) {
        uint8_t const wl = src[pos++];
        uint32_t const windowLog = (wl >> 3) + 10;
        if (windowLog > kWindowLogMax) return mkerr(E_frameParameter_windowTooLarge);
        h.windowSize = 1ULL << windowLog;
        h.windowSize += (h.windowSize >> 3) * (wl & 7);
    }
    switch (dictIDSizeCode) {
        case 1: h.dictID = src[pos]; pos += 1; break;
        case 2: h.dictID = rd16(src + pos); pos += 2; break;
        case 3: h.dictID = rd32(src + pos); pos += 4; break;
        default: break;
    }
    switch (fcsID) {
        case 0: if (singleSegment) h.contentSize = src[pos]; break;
        case 1: h.contentSize = rd16(src + pos) + 256; break;
        case 2: h.contentSize = rd32(src + pos); break;
        case 3: h.contentSize = rd64(src + pos); break;
    }
    if (singleSegment) h.windowSize = h.contentSize;
    return 0;
}

// readSkippableFrameSize (zstd_decompressf.pas:441-459)
size_t skippable_size(const uint8_t* src, size_t srcSize)
{
    if (srcSize < 8) return mkerr(E_srcSize_wrong);
    uint32_t const sz = rd32(src + 4);
    if ((uint32_t)(sz + 8) < sz) return mkerr(E_frameParameter_unsupported);
    size_t const total = (size_t)8 + sz;
    if (total > srcSize) return mkerr(E_srcSize_wrong);
    return total;
}

// Walk one frame: header, then every block header (ZSTD_decompressFrame :734-808 without the decoding,
// i.e. the loop of ZSTD_findFrameSizeInfo :566-639).  Appends block descriptors; returns bytes consumed.
// opts: kOptMagicless (frame format), kOptIgnoreChecksum (the trailer is skipped, not compared: :815-828)
size_t walk_frame(const uint8_t* base, const zstdgpu_frame_in& in, uint32_t frameIdx, uint32_t batchDictID,
                  zstdgpu_frame_desc& fd, std::vector<zstdgpu_block_desc>* blocks, size_t& nBlocksOut, uint32_t opts = 0)
{
    bool const magicless = (opts & kOptMagicless) != 0;
    size_t const pre = magicless ? 1 : 5;
    const uint8_t* const src = base + in.srcOff;
    size_t remaining = in.srcSize;
    fd = zstdgpu_frame_desc{};
    fd.srcOff = in.srcOff; fd.srcSize = in.srcSize; fd.dstOff = in.dstOff; fd.dstCapacity = in.dstCapacity;
    fd.contentSize = ZSTDGPU_CONTENTSIZE_UNKNOWN;
    fd.firstBlock = (uint32_t)nBlocksOut;
    auto fail = [&](uint32_t code) { fd.hostError = code; fd.nBlocks = 0; nBlocksOut = fd.firstBlock; if (blocks) blocks->resize(fd.firstBlock); return (size_t)0; };
    if (remaining < pre + 1 + 3) return fail(E_srcSize_wrong);       // ZSTD_FRAMEHEADERSIZE_MIN(format) + block header (:755)
    size_t const fhsize = frame_header_size(src, pre, magicless);
    if (remaining < fhsize + 3) return fail(E_srcSize_wrong);
    FrameHeader fh;
    size_t const r = get_frame_header(fh, src, fhsize, magicless);
    if (iserr(r)) return fail((uint32_t)(0 - r));
    if (r > 0) return fail(E_srcSize_wrong);
    if (fh.dictID && fh.dictID != batchDictID) return fail(E_dictionary_wrong);   // ZSTD_decodeFrameHeader :541-563
    fd.contentSize = fh.contentSize; fd.windowSize = fh.windowSize; fd.dictID = fh.dictID;
    size_t ip = fhsize;
    remaining -= fhsize;
    for (;;) {
        zstdgpu_block_desc bd{};
        bd.frameIdx = frameIdx;
        auto stop = [&](uint32_t code) { bd.blockType = BT_STOP; bd.hostError = (uint16_t)code; bd.lastBlock = 1; };
        bool done = false;
        if (remaining < 3) { stop(E_srcSize_wrong); done = true; }   // ZSTD_getcBlockSize (zstd_decompress_block.pas:190-207)
        else {
            uint32_t const hdr = rd24(src + ip);
            uint32_t const type = (hdr >> 1) & 3, size = hdr >> 3;
            uint32_t const cSize = type == BT_RLE ? 1 : size;
            if (type == 3) { stop(E_corruption); done = true; }
            else if (cSize > remaining - 3) { stop(E_srcSize_wrong); done = true; }
            else {
                bd.srcOff = in.srcOff + ip + 3;
                bd.cSize = cSize; bd.origSize = size; bd.blockType = (uint8_t)type; bd.lastBlock = hdr & 1;
                ip += 3 + cSize; remaining -= 3 + cSize;
                done = bd.lastBlock;
            }
        }
        if (blocks) blocks->push_back(bd);
        nBlocksOut++;
        if (done) { if (bd.blockType == BT_STOP) { fd.nBlocks = (uint32_t)nBlocksOut - fd.firstBlock; return ip; } break; }
    }
    fd.nBlocks = (uint32_t)nBlocksOut - fd.firstBlock;
    if (fh.checksumFlag) {
        if (remaining < 4) fd.checksumFlag = 2;   // trailer missing: checksum_wrong once everything else passed (:817)
        else { fd.checksumFlag = (opts & kOptIgnoreChecksum) ? 0 : 1; fd.checksum = rd32(src + ip); ip += 4; remaining -= 4; }
    }
    return ip;
}

}  // namespace

// ===================================================================== objects (context and dictionary: zstdgpu_objects.h)
struct zstdgpu_batch {
    zstdgpu_ctx* ctx = nullptr;      // null once the context is gone: the batch still decodes and frees what it holds
    int device = 0;
    const zstdgpu_ddict* ddict = nullptr;
    std::vector<DevBlock> hBlocks;
    std::vector<DevFrame> hFrames;
    std::vector<uint32_t> hLitJobs, hSeqJobs;
    // counts of the arrays above; a batch built on the device fills the host copies only when a query needs them
    uint32_t nFrames = 0, nBlocks = 0, nLitJobs = 0, nSeqJobs = 0;
    bool mirrored = true;
    DevBlock* dBlocks = nullptr; DevBlockState* dState = nullptr; DevFrame* dFrames = nullptr; DevFrameStatus* dStatus = nullptr;
    uint8_t* dLitBuf = nullptr; uint8_t* dLitTmp = nullptr; uint4* dPackTab = nullptr; uint32_t* dSeqBuf = nullptr; uint32_t *dLitJobs = nullptr, *dSeqJobs = nullptr, *dDryList = nullptr;
    size_t litBufBytes = 0, litTmpBytes = 0, seqBufBytes = 0;      // real sizes of the two big buffers (they return to the context's spare list)
    uint64_t litBytes = 0, nSeqTotal = 0, srcBytes = 0, scratchBytes = 0;
    bool seqTablesFromDict = false;   // every sequence job takes its three tables from the dictionary
    bool litDictSingleStream = false; // every literal job is a single stream coded with the dictionary's table
    bool profiling = false;
    hipEvent_t ev[ZSTDGPU_NUM_STAGES + 1] = {};
    bool evReady = false, evRecorded = false;
    hipStream_t lastStream = nullptr;
    bool launched = false;
    // the literal pack kernel runs beside the sequence kernel on a stream of its own, forked and joined with events
    hipStream_t side = nullptr;
    hipEvent_t evFork = nullptr, evJoin = nullptr;
    // recorded behind the last kernel of every decode: what the queries and the destructor wait for (a stream handle kept
    // from the last launch may be gone by then), and what a decode on another stream waits for before it reuses the scratch
    hipEvent_t evDone = nullptr;
};

namespace {

bool set_device(const zstdgpu_ctx* ctx) { return hipSetDevice(ctx->device) == hipSuccess; }
bool set_device(const zstdgpu_batch* b);

template <typename T>
bool dalloc(T*& p, size_t n, uint64_t& acct)
{
    size_t const bytes = (n ? n : 1) * sizeof(T);
    if (hipMalloc((void**)&p, bytes) != hipSuccess) { p = nullptr; return false; }
    acct += bytes;
    return true;
}

// A batch buffer: a spare one of the context when one fits (at least `bytes`, at most twice that), else a new one.
// `got` receives the buffer's real size, which goes back to the context with it.
template <typename T>
bool balloc(zstdgpu_ctx* ctx, T*& p, size_t n, size_t& got, uint64_t& acct)
{
    size_t const bytes = (((n ? n : 1) * sizeof(T)) + 255) & ~(size_t)255;
    int best = -1;
    for (size_t i = 0; i < ctx->spare.size(); i++)
        if (ctx->spare[i].bytes >= bytes && ctx->spare[i].bytes <= 2 * bytes && (best < 0 || ctx->spare[i].bytes < ctx->spare[(size_t)best].bytes)) best = (int)i;
    if (best >= 0) {
        p = (T*)ctx->spare[(size_t)best].p; got = ctx->spare[(size_t)best].bytes;
        ctx->spare.erase(ctx->spare.begin() + best);
    } else {
        if (hipMalloc((void**)&p, bytes) != hipSuccess) { p = nullptr; got = 0; return false; }
        got = bytes;
    }
    acct += got;
    return true;
}
void bfree(zstdgpu_ctx* ctx, void* p, size_t bytes)
{
    if (!p) return;
    if (ctx && bytes >= (1u << 20) && ctx->spare.size() < 16) ctx->spare.push_back({p, bytes});    // small buffers are cheap to allocate again
    else hipFree(p);
}

// Literals-section header (switch of ZSTD_decodeLiteralsBlock, zstd_decompress_block.pas:243-274, :333-400)
// and the count/modes bytes of ZSTD_decodeSeqHeaders (:708-742).
void parse_sections(const uint8_t* payload, DevBlock& d)
{
    uint32_t const cSize = d.cSize;
    if (cSize >= kBlockSizeMax) { d.hostLitErr = E_srcSize_wrong; return; }       // :1513
    if (cSize < 3) { d.hostLitErr = E_corruption; return; }                        // MIN_CBLOCK_SIZE
    uint32_t const type = payload[0] & 3, lhl = (payload[0] >> 2) & 3;
    d.litType = (uint8_t)type;
    uint32_t section = 0;
    if (type >= SET_COMPRESSED) {
        if (type == SET_REPEAT && d.hufSrc == SRC_NONE) { d.hostLitErr = E_dictionary_corrupted; return; }   // :233-234
        if (cSize < 5) { d.hostLitErr = E_corruption; return; }
        uint32_t const lhc = rd32(payload);
        uint32_t lhSize, litSize, litCSize;
        switch (lhl) {
            case 0: case 1: default: d.singleStream = !lhl; lhSize = 3; litSize = (lhc >> 4) & 0x3FF; litCSize = (lhc >> 14) & 0x3FF; break;
            case 2: lhSize = 4; litSize = (lhc >> 4) & 0x3FFF; litCSize = lhc >> 18; break;
            case 3: lhSize = 5; litSize = (lhc >> 4) & 0x3FFFF; litCSize = (lhc >> 22) + ((uint32_t)payload[4] << 10); break;
        }
        if (litSize > kBlockSizeMax) { d.hostLitErr = E_corruption; return; }
        if (litCSize + lhSize > cSize) { d.hostLitErr = E_corruption; return; }
        d.litHdrSize = (uint8_t)lhSize; d.litSize = litSize; d.litCSize = litCSize;
        section = lhSize + litCSize;
    } else {
        uint32_t lhSize, litSize;
        switch (lhl) {
            case 0: case 2: default: lhSize = 1; litSize = payload[0] >> 3; break;
            case 1: lhSize = 2; litSize = rd16(payload) >> 4; break;
            case 3: lhSize = 3; litSize = rd24(payload) >> 4;
                    if (type == SET_RLE && cSize < 4) { d.hostLitErr = E_corruption; return; }
                    break;
        }
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
    const uint8_t* ip = payload + section;
    const uint8_t* const iend = payload + cSize;
    if (rem < 1) { d.hostSeqErr = E_srcSize_wrong; return; }
    uint32_t nbSeq = *ip++;
    if (!nbSeq) { if (rem != 1) d.hostSeqErr = E_srcSize_wrong; d.nbSeq = 0; return; }
    if (nbSeq > 0x7F) {
        if (nbSeq == 0xFF) { if (ip + 2 > iend) { d.hostSeqErr = E_srcSize_wrong; return; } nbSeq = rd16(ip) + kLongNbSeq; ip += 2; }
        else { if (ip >= iend) { d.hostSeqErr = E_srcSize_wrong; return; } nbSeq = ((nbSeq - 0x80) << 8) + *ip++; }
    }
    d.nbSeq = nbSeq;
    if (ip + 1 > iend) { d.hostSeqErr = E_srcSize_wrong; d.nbSeq = 0; return; }
    d.llType = *ip >> 6; d.ofType = (*ip >> 4) & 3; d.mlType = (*ip >> 2) & 3;
    ip++;
    d.seqHdrOff = (uint32_t)(ip - payload);
}

bool set_device(const zstdgpu_batch* b) { return hipSetDevice(b->device) == hipSuccess; }

// Every launch of the batch so far is over (launches on different streams are chained through evDone, so the last one
// covers them all).  An event that cannot be waited for (recorded under stream capture) falls back to the whole device.
bool wait_done(const zstdgpu_batch* b)
{
    if (!b->launched) return true;
    if (b->evDone && hipEventSynchronize(b->evDone) == hipSuccess) return true;
    (void)hipGetLastError();
    return hipDeviceSynchronize() == hipSuccess;
}

// A batch is known to its context so that a context destroyed first can tell it (zstdgpu_destroy).
bool adopt_batch(zstdgpu_ctx* ctx, zstdgpu_batch* b)
{
    b->ctx = ctx; b->device = ctx->device;
    try { ctx->batches.push_back(b); } catch (...) { b->ctx = nullptr; return false; }
    return true;
}
void drop_batch(zstdgpu_batch* b)
{
    if (b->ctx) {
        auto& v = b->ctx->batches;
        for (size_t i = 0; i < v.size(); i++) if (v[i] == b) { v.erase(v.begin() + (long)i); break; }
    }
    delete b;
}

void free_batch_device(zstdgpu_batch* b)
{
    // zstdgpu_batch_decode is asynchronous and nothing obliges the caller to wait before destroying the batch: the
    // big buffers go back to the context's spare list (no implicit synchronisation as with hipFree), where the next
    // batch may pick them up and launch on another stream -- so the last launch must be over first.
    wait_done(b);
    if (b->launched && b->side) hipStreamSynchronize(b->side);
    hipFree(b->dBlocks); hipFree(b->dState); hipFree(b->dFrames); hipFree(b->dStatus);
    bfree(b->ctx, b->dLitBuf, b->litBufBytes); bfree(b->ctx, b->dLitTmp, b->litTmpBytes); bfree(b->ctx, b->dSeqBuf, b->seqBufBytes);
    hipFree(b->dLitJobs); hipFree(b->dSeqJobs); hipFree(b->dPackTab); hipFree(b->dDryList);
    if (b->evReady) for (auto& e : b->ev) hipEventDestroy(e);
    if (b->side) hipStreamDestroy(b->side);
    if (b->evFork) hipEventDestroy(b->evFork);
    if (b->evJoin) hipEventDestroy(b->evJoin);
    if (b->evDone) hipEventDestroy(b->evDone);
}

}  // namespace

// ===================================================================== error helpers
extern "C" unsigned zstdgpu_isError(size_t code) { return iserr(code); }
extern "C" unsigned zstdgpu_getErrorCode(size_t code) { return iserr(code) ? (unsigned)((size_t)0 - code) : 0; }
extern "C" unsigned zstdgpu_versionNumber(void) { return ZSTDGPU_VERSION_NUMBER; }
extern "C" const char* zstdgpu_getErrorName(size_t code)
{
    switch (zstdgpu_getErrorCode(code)) {   // strings of ERR_getErrorString (error_private.pas:70-106)
        case 0: return "No error detected";
        case 1: return "Error (generic)";
        case 10: return "Unknown frame descriptor";
        case 12: return "Version not supported";
        case 14: return "Unsupported frame parameter";
        case 16: return "Frame requires too much memory for decoding";
        case 20: return "Corrupted block detected";
        case 22: return "Restored data doesn't match checksum";
        case 30: return "Dictionary is corrupted";
        case 32: return "Dictionary mismatch";
        case 40: return "Unsupported parameter";
        case 42: return "Parameter is out of bound";
        case 60: return "Operation not authorized at current processing stage";
        case 104: return "Destination buffer is wrong";
        case 64: return "Allocation error : not enough memory";
        case 70: return "Destination buffer is too small";
        case 72: return "Src size is incorrect";
        default: return "Unspecified error code";
    }
}

// ===================================================================== context
static void stream_detach(zstdgpu_dstream* z);      // a stream whose context is gone: every later call fails
static void bufferless_free(zstdgpu_bufferless* b);

extern "C" zstdgpu_ctx* zstdgpu_create(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return nullptr;
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    zstdgpu_ctx* c = new (std::nothrow) zstdgpu_ctx;
    if (c) { c->device = device; c->scratch = noalloc_scratch(device); }
    return c;
}
extern "C" void zstdgpu_destroy(zstdgpu_ctx* ctx)
{
    if (!ctx) return;
    set_device(ctx);
    // dictionaries and streams that outlive the context keep working as far as they can: they forget it
    for (auto* z : ctx->streams) stream_detach(z);
    for (auto* d : ctx->ddicts) d->ctx = nullptr;
    for (auto* b : ctx->batches) b->ctx = nullptr;
    bufferless_free(ctx->bufferless);
    hipFree(ctx->dSrc); hipFree(ctx->dDst); hipFree(ctx->oneShotScratch); hipFree(ctx->oneShotIO);
    for (auto& sp : ctx->spare) hipFree(sp.p);
    delete ctx;
}

// ===================================================================== header queries
namespace {
unsigned long long frame_content_size(const uint8_t* s, size_t srcSize, bool magicless)
{
    // ZSTD_getFrameContentSize (:414-431) over ZSTD_getFrameHeader_advanced (:296-318): a skippable frame needs its
    // 8-byte header to be there and has content size 0
    if (!magicless && srcSize >= 5 && (rd32(s) & kMagicSkipMask) == kMagicSkipStart) return srcSize >= 8 ? 0 : ZSTDGPU_CONTENTSIZE_ERROR;
    FrameHeader h;
    if (get_frame_header(h, s, srcSize, magicless) != 0) return ZSTDGPU_CONTENTSIZE_ERROR;
    return h.contentSize;
}
size_t find_frame_size(const uint8_t* src, size_t srcSize, bool magicless);
}  // namespace
extern "C" unsigned long long zstdgpu_getFrameContentSize(const void* src, size_t srcSize)
{
    return frame_content_size((const uint8_t*)src, srcSize, false);
}
extern "C" size_t zstdgpu_findFrameCompressedSize(const void* srcv, size_t srcSize)
{
    return find_frame_size((const uint8_t*)srcv, srcSize, false);
}
namespace {
size_t find_frame_size(const uint8_t* src, size_t srcSize, bool magicless)
{
    if (srcSize >= 8 && (rd32(src) & kMagicSkipMask) == kMagicSkipStart) return skippable_size(src, srcSize);
    if (!magicless && srcSize >= 5 && (rd32(src) & kMagicSkipMask) == kMagicSkipStart) return mkerr(E_srcSize_wrong);   // half a skippable header (:303-308, :599)
    FrameHeader h;
    size_t const r = get_frame_header(h, src, srcSize, magicless);
    if (iserr(r)) return r;
    if (r > 0) return mkerr(E_srcSize_wrong);
    size_t ip = h.headerSize, remaining = srcSize - h.headerSize;
    for (;;) {
        if (remaining < 3) return mkerr(E_srcSize_wrong);
        uint32_t const hdr = rd24(src + ip);
        uint32_t const type = (hdr >> 1) & 3;
        if (type == 3) return mkerr(E_corruption);
        size_t const cSize = type == BT_RLE ? 1 : (hdr >> 3);
        if (3 + cSize > remaining) return mkerr(E_srcSize_wrong);
        ip += 3 + cSize; remaining -= 3 + cSize;
        if (hdr & 1) break;
    }
    if (h.checksumFlag) { if (remaining < 4) return mkerr(E_srcSize_wrong); ip += 4; }
    return ip;
}
}  // namespace
// ZSTD_isFrame (zstd_decompressf.pas:223-239)
extern "C" unsigned zstdgpu_isFrame(const void* src, size_t srcSize)
{
    if (srcSize < 4) return 0;
    uint32_t const magic = rd32((const uint8_t*)src);
    return magic == kMagic || (magic & kMagicSkipMask) == kMagicSkipStart;
}
// ZSTD_frameHeaderSize (:269-272)
extern "C" size_t zstdgpu_frameHeaderSize(const void* src, size_t srcSize)
{
    return frame_header_size((const uint8_t*)src, srcSize, false);
}
// ZSTD_getFrameHeader (:403-406): 0, or the number of bytes wanted, or an error
extern "C" size_t zstdgpu_getFrameHeader(zstdgpu_frameHeader* zfh, const void* srcv, size_t srcSize)
{
    const uint8_t* const src = (const uint8_t*)srcv;
    if (!zfh) return mkerr(E_generic);
    memset(zfh, 0, sizeof(*zfh));
    if (srcSize < 5) return 5;
    if (!src) return mkerr(E_generic);
    if ((rd32(src) & kMagicSkipMask) == kMagicSkipStart) {      // :303-313
        if (srcSize < 8) return 8;
        zfh->frameContentSize = rd32(src + 4);
        zfh->frameType = 1;
        return 0;
    }
    FrameHeader h;
    size_t const r = get_frame_header(h, src, srcSize, false);
    if (r != 0) return r;
    zfh->frameContentSize = h.contentSize; zfh->windowSize = h.windowSize;
    zfh->blockSizeMax = (unsigned)(h.windowSize < kBlockSizeMax ? h.windowSize : kBlockSizeMax);     // :389
    zfh->frameType = 0; zfh->headerSize = h.headerSize; zfh->dictID = h.dictID; zfh->checksumFlag = h.checksumFlag;
    return 0;
}
// ZSTD_getDecompressedSize (:525-535)
extern "C" unsigned long long zstdgpu_getDecompressedSize(const void* src, size_t srcSize)
{
    unsigned long long const r = zstdgpu_getFrameContentSize(src, srcSize);
    return r >= ZSTDGPU_CONTENTSIZE_ERROR ? 0 : r;
}
// ZSTD_findDecompressedSize (:466-515): sum of the content sizes of all the frames in src
extern "C" unsigned long long zstdgpu_findDecompressedSize(const void* srcv, size_t srcSize)
{
    const uint8_t* src = (const uint8_t*)srcv;
    unsigned long long total = 0;
    while (srcSize >= 5) {
        if ((rd32(src) & kMagicSkipMask) == kMagicSkipStart) {
            size_t const sk = skippable_size(src, srcSize);
            if (iserr(sk)) return ZSTDGPU_CONTENTSIZE_ERROR;
            src += sk; srcSize -= sk;
            continue;
        }
        unsigned long long const r = zstdgpu_getFrameContentSize(src, srcSize);
        if (r >= ZSTDGPU_CONTENTSIZE_ERROR) return r;
        if (total + r < total) return ZSTDGPU_CONTENTSIZE_ERROR;
        total += r;
        size_t const fsz = find_frame_size(src, srcSize, false);
        if (iserr(fsz)) return ZSTDGPU_CONTENTSIZE_ERROR;
        src += fsz; srcSize -= fsz;
    }
    if (srcSize) return ZSTDGPU_CONTENTSIZE_ERROR;
    return total;
}
// ZSTD_decompressBound (:660-681) over ZSTD_findFrameSizeInfo (:566-639): content size where the header has it, else
// blocks x blockSizeMax; skippable frames add nothing
extern "C" unsigned long long zstdgpu_decompressBound(const void* srcv, size_t srcSize)
{
    const uint8_t* src = (const uint8_t*)srcv;
    unsigned long long bound = 0;
    while (srcSize > 0) {
        if (srcSize >= 8 && (rd32(src) & kMagicSkipMask) == kMagicSkipStart) {
            size_t const sk = skippable_size(src, srcSize);
            if (iserr(sk)) return ZSTDGPU_CONTENTSIZE_ERROR;
            src += sk; srcSize -= sk;
            continue;
        }
        FrameHeader h;
        if (get_frame_header(h, src, srcSize, false) != 0) return ZSTDGPU_CONTENTSIZE_ERROR;
        size_t ip = h.headerSize, remaining = srcSize - h.headerSize;
        unsigned long long nbBlocks = 0;
        for (;;) {
            if (remaining < 3) return ZSTDGPU_CONTENTSIZE_ERROR;
            uint32_t const hdr = rd24(src + ip);
            uint32_t const type = (hdr >> 1) & 3;
            if (type == 3) return ZSTDGPU_CONTENTSIZE_ERROR;
            size_t const cSize = type == BT_RLE ? 1 : (hdr >> 3);
            if (3 + cSize > remaining) return ZSTDGPU_CONTENTSIZE_ERROR;
            ip += 3 + cSize; remaining -= 3 + cSize;
            nbBlocks++;
            if (hdr & 1) break;
        }
        if (h.checksumFlag) { if (remaining < 4) return ZSTDGPU_CONTENTSIZE_ERROR; ip += 4; }
        unsigned long long const blockMax = h.windowSize < kBlockSizeMax ? h.windowSize : kBlockSizeMax;
        bound += h.contentSize != ZSTDGPU_CONTENTSIZE_UNKNOWN ? h.contentSize : nbBlocks * blockMax;
        src += ip; srcSize -= ip;
    }
    return bound;
}
extern "C" unsigned zstdgpu_getDictID_fromDict(const void* dict, size_t dictSize)
{
    if (dictSize < 8 || rd32((const uint8_t*)dict) != kMagicDict) return 0;
    return rd32((const uint8_t*)dict + 4);
}
extern "C" unsigned zstdgpu_getDictID_fromFrame(const void* src, size_t srcSize)
{
    FrameHeader h;
    if (get_frame_header(h, (const uint8_t*)src, srcSize) != 0) return 0;
    return h.dictID;
}

// ===================================================================== dictionaries
extern "C" zstdgpu_ddict* zstdgpu_createDDict(zstdgpu_ctx* ctx, const void* dict, size_t dictSize)
{
    if (!ctx || !set_device(ctx) || dictSize > 0x7FFFFFFFu) return nullptr;
    zstdgpu_ddict* d = new (std::nothrow) zstdgpu_ddict;
    if (!d) return nullptr;
    d->ctx = ctx; d->size = (uint32_t)dictSize; d->device = ctx->device;
    try { ctx->ddicts.push_back(d); } catch (...) { delete d; return nullptr; }
    uint64_t acct = 0;
    if (!dalloc(d->dBytes, dictSize + 16, acct) || !dalloc(d->dDict, 1, acct)) { zstdgpu_freeDDict(d); return nullptr; }
    if (dictSize && hipMemcpy(d->dBytes, dict, dictSize, hipMemcpyHostToDevice) != hipSuccess) { zstdgpu_freeDDict(d); return nullptr; }
    // ZSTD_decompress_insertDictionary (zstd_decompressf.pas:1358-1384): magic -> entropy tables + content, else raw content
    if (dictSize >= 8 && rd32((const uint8_t*)dict) == kMagicDict) {
        d->hasEntropy = 1; d->entropyMask = 3;
        d->dictID = rd32((const uint8_t*)dict + 4);
        launch_digest_dict(d->dBytes, d->size, d->dDict, nullptr);
        uint32_t head[3] = {0, 0, 0};   // status, hasEntropy, entropySize
        if (hipMemcpy(head, d->dDict, sizeof(head), hipMemcpyDeviceToHost) != hipSuccess) { zstdgpu_freeDDict(d); return nullptr; }
        d->status = head[0]; d->entropySize = head[2];
    } else {
        DevDict* h = new (std::nothrow) DevDict;
        if (!h) { zstdgpu_freeDDict(d); return nullptr; }
        memset(h, 0, sizeof(*h));
        hipError_t const e = hipMemcpy(d->dDict, h, sizeof(*h), hipMemcpyHostToDevice);
        delete h;
        if (e != hipSuccess) { zstdgpu_freeDDict(d); return nullptr; }
    }
    return d;
}
extern "C" void zstdgpu_freeDDict(zstdgpu_ddict* d)
{
    if (!d) return;
    if (d->ctx) {
        if (d->ctx->refDDict == d) d->ctx->refDDict = nullptr;   // a freed dictionary is no longer referenced
        auto& v = d->ctx->ddicts;
        v.erase(std::remove(v.begin(), v.end(), d), v.end());
    }
    hipSetDevice(d->device);
    hipFree(d->dBytes); hipFree(d->dDict);
    delete d;
}

// ===================================================================== descriptor builder
namespace {
size_t build_descriptors(const void* hostSrc, size_t srcBytes, const zstdgpu_frame_in* in, uint32_t nFrames,
                         zstdgpu_frame_desc* frames, zstdgpu_block_desc* blocks, size_t blockCapacity, uint32_t opts)
try {
    // dictID checks need the batch dictionary: done again in zstdgpu_batch_create; here any dictID is accepted
    std::vector<zstdgpu_block_desc> tmp;
    size_t nBlocks = 0;
    for (uint32_t i = 0; i < nFrames; i++) {
        zstdgpu_frame_desc fd;
        if (in[i].srcOff > srcBytes || in[i].srcSize > srcBytes - in[i].srcOff) {
            fd = zstdgpu_frame_desc{}; fd.srcOff = in[i].srcOff; fd.srcSize = in[i].srcSize; fd.dstOff = in[i].dstOff; fd.dstCapacity = in[i].dstCapacity;
            fd.firstBlock = (uint32_t)nBlocks; fd.hostError = E_srcSize_wrong; fd.contentSize = ZSTDGPU_CONTENTSIZE_UNKNOWN;
        } else {
            // peek the dictID so the walk does not reject it; the batch compares it with its dictionary
            FrameHeader fh;
            uint32_t did = 0;
            if (get_frame_header(fh, (const uint8_t*)hostSrc + in[i].srcOff, in[i].srcSize, (opts & kOptMagicless) != 0) == 0) did = fh.dictID;
            size_t const used = walk_frame((const uint8_t*)hostSrc, in[i], i, did, fd, &tmp, nBlocks, opts);
            if (!fd.hostError && used != in[i].srcSize && fd.checksumFlag != 2 && !(fd.nBlocks && tmp.back().blockType == BT_STOP))
                { fd.hostError = E_srcSize_wrong; }   // bytes after the frame: one entry is exactly one frame
        }
        if (frames) frames[i] = fd;
    }
    if (blocks) {
        if (nBlocks > blockCapacity) return mkerr(E_dstSize_tooSmall);
        if (nBlocks) memcpy(blocks, tmp.data(), nBlocks * sizeof(zstdgpu_block_desc));
    }
    return nBlocks;
}
catch (...) { return mkerr(E_memory_allocation); }      // the C ABI never throws (std::bad_alloc from the host containers)
}  // namespace
extern "C" size_t zstdgpu_buildDescriptors(const void* hostSrc, size_t srcBytes, const zstdgpu_frame_in* in, uint32_t nFrames,
                                           zstdgpu_frame_desc* frames, zstdgpu_block_desc* blocks, size_t blockCapacity)
{
    return build_descriptors(hostSrc, srcBytes, in, nFrames, frames, blocks, blockCapacity, 0);
}

// ===================================================================== batch
extern "C" zstdgpu_batch* zstdgpu_batch_create(zstdgpu_ctx* ctx, const void* hostSrc, size_t srcBytes,
                                               const zstdgpu_frame_desc* frames, uint32_t nFrames,
                                               const zstdgpu_block_desc* blocks, uint32_t nBlocks, const zstdgpu_ddict* ddict)
{
    if (!ctx || !set_device(ctx)) return nullptr;
    zstdgpu_batch* B = new (std::nothrow) zstdgpu_batch;
    if (!B) return nullptr;
    if (!adopt_batch(ctx, B)) { delete B; return nullptr; }
    B->ddict = ddict;
    const uint8_t* const src = (const uint8_t*)hostSrc;
    bool const dictEntropy = ddict && ddict->hasEntropy;
    uint32_t const entropyMask = ddict ? ddict->entropyMask : 0u;      // bit 0 Huffman table, bit 1 sequence tables
    uint32_t const batchDictID = ddict ? ddict->dictID : 0;
    uint64_t litOff = 0, seqOff = 0, tmpOff = 0;
    uint32_t ownTables = 0;      // sequence jobs with a table that is not the dictionary's
    uint32_t otherLits = 0;      // literal jobs that are not a single stream coded with the dictionary's table
    try {
    B->hBlocks.resize(nBlocks);
    B->hFrames.resize(nFrames);
    for (uint32_t fi = 0; fi < nFrames; fi++) {
        const zstdgpu_frame_desc& fd = frames[fi];
        DevFrame& f = B->hFrames[fi];
        f.dstOff = fd.dstOff; f.dstCapacity = fd.dstCapacity; f.contentSize = fd.contentSize;
        f.firstBlock = fd.firstBlock; f.nBlocks = fd.nBlocks;
        f.checksumFlag = fd.checksumFlag; f.checksum = fd.checksum;
        f.hostError = fd.hostError;
        f.useDict = ddict ? 1 : 0;
        f.longWindow = fd.windowSize > (1ull << 24) ? 1 : 0;
        if (!f.hostError && fd.dictID && fd.dictID != batchDictID) f.hostError = E_dictionary_wrong;
        if (!f.hostError && (uint64_t)fd.firstBlock + fd.nBlocks > nBlocks) f.hostError = E_generic;
        if (!f.hostError && fd.dstCapacity > 0x7FFFFFFEull) f.dstCapacity = 0x7FFFFFFEull;   // frames regenerate < 2 GiB (the port's sizes are int32)
        if (f.hostError) { f.nBlocks = 0; f.longWindow = 0; continue; }
        B->srcBytes += fd.srcSize;
        // entropy state carried from block to block inside the frame (zstd_ddict.pas:109-126): which block defines each table
        int32_t hufDef = (entropyMask & 1u) ? SRC_DICT : SRC_NONE;
        int32_t llDef = (entropyMask & 2u) ? SRC_DICT : SRC_NONE, ofDef = llDef, mlDef = llDef;
        bool dead = false;   // a host-detected error ends the frame: later blocks are never reached
        for (uint32_t b = fd.firstBlock; b < fd.firstBlock + fd.nBlocks; b++) {
            const zstdgpu_block_desc& bd = blocks[b];
            DevBlock& d = B->hBlocks[b];
            d = DevBlock{};
            d.srcOff = bd.srcOff; d.cSize = bd.cSize; d.origSize = bd.origSize; d.frameIdx = fi;
            d.blockType = bd.blockType; d.lastBlock = bd.lastBlock;
            d.hufSrc = d.llSrc = d.ofSrc = d.mlSrc = SRC_NONE;
            if (bd.blockType == BT_STOP) { d.hostLitErr = bd.hostError ? bd.hostError : (uint16_t)E_corruption; dead = true; continue; }
            if (dead) { d.blockType = BT_STOP; d.hostLitErr = E_generic; continue; }
            if (bd.srcOff > srcBytes || bd.cSize > srcBytes - bd.srcOff) { d.blockType = BT_STOP; d.hostLitErr = E_srcSize_wrong; dead = true; continue; }
            if (bd.blockType != BT_COMPRESSED) continue;
            d.hufSrc = hufDef;
            parse_sections(src + bd.srcOff, d);
            if (d.hostLitErr) { dead = true; continue; }
            if (d.litType == SET_COMPRESSED) d.hufSrc = hufDef = (int32_t)b;
            if (d.litType >= SET_COMPRESSED) {
                d.litBufOff = litOff;
                litOff += (d.litSize + 15) & ~15ull;
                if (!d.singleStream) { d.litTmpOff = tmpOff; tmpOff += litTmpBytes(d.litSize); }
                if (!d.singleStream || d.hufSrc != SRC_DICT) otherLits++;
                B->hLitJobs.push_back(b);
            }
            if (d.hostSeqErr) { dead = true; continue; }
            if (d.nbSeq) {
                if (d.llType != SET_REPEAT) llDef = (int32_t)b;
                if (d.ofType != SET_REPEAT) ofDef = (int32_t)b;
                if (d.mlType != SET_REPEAT) mlDef = (int32_t)b;
                d.llSrc = llDef; d.ofSrc = ofDef; d.mlSrc = mlDef;
                if (llDef != SRC_DICT || ofDef != SRC_DICT || mlDef != SRC_DICT) ownTables++;
                d.seqBufOff = seqOff;
                seqOff += (uint64_t)d.nbSeq + 1 + 64;   // + the terminal record + 64 spare records (the sequence kernel stores whole rounds)
                B->hSeqJobs.push_back(b);
            }
        }
    }
    } catch (...) { drop_batch(B); return nullptr; }      // the C ABI never throws
    B->litBytes = litOff + 64; B->nSeqTotal = seqOff;
    B->nFrames = nFrames; B->nBlocks = nBlocks; B->nLitJobs = (uint32_t)B->hLitJobs.size(); B->nSeqJobs = (uint32_t)B->hSeqJobs.size();
    B->seqTablesFromDict = dictEntropy && B->nSeqJobs != 0 && ownTables == 0;
    B->litDictSingleStream = dictEntropy && B->nLitJobs != 0 && otherLits == 0;
    bool ok = dalloc(B->dBlocks, nBlocks, B->scratchBytes) && dalloc(B->dState, nBlocks, B->scratchBytes) &&
              dalloc(B->dFrames, nFrames, B->scratchBytes) && dalloc(B->dStatus, nFrames, B->scratchBytes) &&
              balloc(ctx, B->dLitBuf, B->litBytes, B->litBufBytes, B->scratchBytes) && balloc(ctx, B->dLitTmp, tmpOff + 64, B->litTmpBytes, B->scratchBytes) && balloc(ctx, B->dSeqBuf, seqOff * 4 + 4, B->seqBufBytes, B->scratchBytes) &&
              dalloc(B->dLitJobs, B->hLitJobs.size(), B->scratchBytes) && dalloc(B->dPackTab, B->hLitJobs.size() * 33 + 1, B->scratchBytes) && dalloc(B->dSeqJobs, B->hSeqJobs.size(), B->scratchBytes) &&
              dalloc(B->dDryList, B->hSeqJobs.size() + 4, B->scratchBytes);
    auto up = [&](void* d, const void* h, size_t n) { return n == 0 || hipMemcpy(d, h, n, hipMemcpyHostToDevice) == hipSuccess; };
    ok = ok && up(B->dBlocks, B->hBlocks.data(), nBlocks * sizeof(DevBlock)) && up(B->dFrames, B->hFrames.data(), nFrames * sizeof(DevFrame)) &&
         up(B->dLitJobs, B->hLitJobs.data(), B->hLitJobs.size() * 4) && up(B->dSeqJobs, B->hSeqJobs.data(), B->hSeqJobs.size() * 4);
    ok = ok && hipStreamCreateWithFlags(&B->side, hipStreamNonBlocking) == hipSuccess &&
         hipEventCreateWithFlags(&B->evFork, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&B->evJoin, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&B->evDone, hipEventDisableTiming) == hipSuccess;
    if (!ok) { free_batch_device(B); drop_batch(B); return nullptr; }
    return B;
}

extern "C" zstdgpu_batch* zstdgpu_batch_create_from_frames(zstdgpu_ctx* ctx, const void* hostSrc, size_t srcBytes,
                                                           const zstdgpu_frame_in* in, uint32_t nFrames, const zstdgpu_ddict* ddict)
try {
    if (!ctx) return nullptr;
    uint32_t const opts = ctx->opts();      // the context's frame format and checksum policy apply to the batch
    std::vector<zstdgpu_frame_desc> frames(nFrames);
    size_t const n = build_descriptors(hostSrc, srcBytes, in, nFrames, frames.data(), nullptr, 0, opts);
    if (iserr(n)) return nullptr;
    std::vector<zstdgpu_block_desc> blocks(n ? n : 1);
    size_t const n2 = build_descriptors(hostSrc, srcBytes, in, nFrames, frames.data(), blocks.data(), blocks.size(), opts);
    if (iserr(n2)) return nullptr;
    return zstdgpu_batch_create(ctx, hostSrc, srcBytes, frames.data(), nFrames, blocks.data(), (uint32_t)n2, ddict);
}
catch (...) { return nullptr; }      // the C ABI never throws (std::bad_alloc from the host containers)

// The same batch, built without a host copy of the source: the frame / block / section headers are walked by
// device kernels over the source in device memory (descriptors.hip); the host only sizes the buffers from the
// totals they report and keeps a mirror of the small descriptor arrays for the query entry points.
extern "C" zstdgpu_batch* zstdgpu_batch_create_on_device(zstdgpu_ctx* ctx, const void* devSrc, size_t srcBytes,
                                                         const zstdgpu_frame_in* in, uint32_t nFrames, const zstdgpu_ddict* ddict)
{
    static_assert(sizeof(zstdgpu_frame_in) == sizeof(FrameIn), "frame entry layout");
    if (!ctx || !set_device(ctx) || (nFrames && (!in || !devSrc))) return nullptr;
    zstdgpu_batch* B = new (std::nothrow) zstdgpu_batch;
    if (!B) return nullptr;
    if (!adopt_batch(ctx, B)) { delete B; return nullptr; }
    B->ddict = ddict;
    uint32_t const batchDictID = ddict ? ddict->dictID : 0;
    FrameIn* dIn = nullptr; uint32_t* dSlots = nullptr; uint64_t* dTotals = nullptr; uint8_t* dFlags = nullptr;
    uint64_t tmpAcct = 0;
    uint64_t totals[16] = {};
    uint32_t nBlocks = 0;
    auto cleanup = [&]() { hipFree(dIn); hipFree(dSlots); hipFree(dTotals); hipFree(dFlags); };
    bool ok = dalloc(B->dFrames, nFrames, B->scratchBytes) && dalloc(B->dStatus, nFrames, B->scratchBytes) &&
              dalloc(dIn, nFrames, tmpAcct) && dalloc(dSlots, nFrames, tmpAcct) && dalloc(dTotals, 16, tmpAcct);
    ok = ok && (nFrames == 0 || hipMemcpy(dIn, in, (size_t)nFrames * sizeof(FrameIn), hipMemcpyHostToDevice) == hipSuccess);
    if (ok) {
        launch_scan_frames((const uint8_t*)devSrc, srcBytes, dIn, nFrames, batchDictID, (ddict ? (uint32_t)kOptDict : 0u) | ctx->opts(), B->dFrames, dSlots, dTotals, nullptr);
        ok = hipMemcpy(totals, dTotals, sizeof(totals), hipMemcpyDeviceToHost) == hipSuccess && totals[0] <= 0x7FFFFFFFull;
        nBlocks = (uint32_t)totals[0];
    }
    ok = ok && dalloc(B->dBlocks, nBlocks, B->scratchBytes) && dalloc(B->dState, nBlocks, B->scratchBytes) && dalloc(dFlags, nBlocks, tmpAcct) &&
         dalloc(B->dLitJobs, nBlocks, B->scratchBytes) && dalloc(B->dSeqJobs, nBlocks, B->scratchBytes);
    ok = ok && hipMemset(B->dBlocks, 0, (size_t)(nBlocks ? nBlocks : 1) * sizeof(DevBlock)) == hipSuccess && hipMemset(dFlags, 0, nBlocks ? nBlocks : 1) == hipSuccess;
    if (ok) {
        launch_fill_blocks((const uint8_t*)devSrc, srcBytes, dIn, nFrames, batchDictID, (ddict ? (uint32_t)kOptDict : 0u) | ctx->opts(), ddict ? ddict->entropyMask : 0u,
                           B->dFrames, B->dBlocks, dFlags, nBlocks, B->dLitJobs, B->dSeqJobs, dTotals, nullptr);
        ok = hipMemcpy(totals, dTotals, sizeof(totals), hipMemcpyDeviceToHost) == hipSuccess;
    }
    if (ok) {
        uint64_t const litOff = totals[1], tmpOff = totals[2], seqOff = totals[3];
        size_t const nLit = (size_t)totals[4], nSeq = (size_t)totals[5];
        B->litBytes = litOff + 64; B->nSeqTotal = seqOff;
        ok = balloc(ctx, B->dLitBuf, B->litBytes, B->litBufBytes, B->scratchBytes) && balloc(ctx, B->dLitTmp, tmpOff + 64, B->litTmpBytes, B->scratchBytes) &&
             balloc(ctx, B->dSeqBuf, seqOff * 4 + 4, B->seqBufBytes, B->scratchBytes) && dalloc(B->dPackTab, nLit * 33 + 1, B->scratchBytes) &&
             dalloc(B->dDryList, nSeq + 4, B->scratchBytes);
        B->nFrames = nFrames; B->nBlocks = nBlocks; B->nLitJobs = (uint32_t)nLit; B->nSeqJobs = (uint32_t)nSeq;
        B->srcBytes = totals[6];
        B->seqTablesFromDict = ddict && ddict->hasEntropy && nSeq != 0 && totals[8] == 0;
        B->litDictSingleStream = ddict && ddict->hasEntropy && nLit != 0 && totals[9] == 0;
        B->mirrored = false;
    }
    ok = ok && hipStreamCreateWithFlags(&B->side, hipStreamNonBlocking) == hipSuccess &&
         hipEventCreateWithFlags(&B->evFork, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&B->evJoin, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&B->evDone, hipEventDisableTiming) == hipSuccess;
    cleanup();
    if (!ok) { free_batch_device(B); drop_batch(B); return nullptr; }
    return B;
}

// Host copies of the descriptor arrays of a batch that was built on the device.
static bool mirror(zstdgpu_batch* B)
try {
    if (B->mirrored) return true;
    if (!set_device(B)) return false;
    B->hBlocks.resize(B->nBlocks); B->hFrames.resize(B->nFrames); B->hLitJobs.resize(B->nLitJobs); B->hSeqJobs.resize(B->nSeqJobs);
    auto down = [&](void* h, const void* d, size_t n) { return n == 0 || hipMemcpy(h, d, n, hipMemcpyDeviceToHost) == hipSuccess; };
    bool const ok = down(B->hBlocks.data(), B->dBlocks, (size_t)B->nBlocks * sizeof(DevBlock)) && down(B->hFrames.data(), B->dFrames, (size_t)B->nFrames * sizeof(DevFrame)) &&
                    down(B->hLitJobs.data(), B->dLitJobs, (size_t)B->nLitJobs * 4) && down(B->hSeqJobs.data(), B->dSeqJobs, (size_t)B->nSeqJobs * 4);
    B->mirrored = ok;
    return ok;
}
catch (...) { return false; }      // the C ABI never throws (std::bad_alloc from the host containers)

// FNV-1a over the descriptor arrays of a batch (blocks, frames, job lists): two batches built from the same input by
// the host walk and by the device walk must agree.
extern "C" uint64_t zstdgpu_batch_descriptorDigest(zstdgpu_batch* b)
{
    if (!b || !mirror(b)) return 0;
    uint64_t h = 0xCBF29CE484222325ull;
    auto mix = [&](const void* p, size_t n) { const uint8_t* q = (const uint8_t*)p; for (size_t i = 0; i < n; i++) { h ^= q[i]; h *= 0x100000001B3ull; } };
    if (!b) return 0;
    mix(b->hBlocks.data(), b->hBlocks.size() * sizeof(DevBlock));
    mix(b->hFrames.data(), b->hFrames.size() * sizeof(DevFrame));
    mix(b->hLitJobs.data(), b->hLitJobs.size() * 4);
    mix(b->hSeqJobs.data(), b->hSeqJobs.size() * 4);
    return h;
}

extern "C" uint32_t zstdgpu_batch_kernelVariants(const zstdgpu_batch* b)
{
    return b ? (b->seqTablesFromDict ? 1u : 0u) | (b->litDictSingleStream ? 2u : 0u) : 0u;
}

extern "C" void zstdgpu_batch_destroy(zstdgpu_batch* b)
{
    if (!b) return;
    set_device(b);
    free_batch_device(b);
    drop_batch(b);
}

extern "C" size_t zstdgpu_batch_decode(zstdgpu_batch* B, const void* devSrc, void* devDst, void* stream)
{
    if (!B || !set_device(B)) return mkerr(E_generic);
    hipStream_t const s = (hipStream_t)stream;
    BatchParams p{};
    p.src = (const uint8_t*)devSrc; p.dst = (uint8_t*)devDst;
    p.blocks = B->dBlocks; p.state = B->dState; p.frames = B->dFrames; p.status = B->dStatus;
    p.litBuf = B->dLitBuf; p.litTmp = B->dLitTmp; p.packTab = B->dPackTab; p.seqBuf = B->dSeqBuf; p.litJobs = B->dLitJobs; p.seqJobs = B->dSeqJobs; p.dryList = B->dDryList;
    p.nLitJobs = B->nLitJobs; p.nSeqJobs = B->nSeqJobs;
    p.nFrames = B->nFrames; p.nBlocks = B->nBlocks;
    p.execMode = B->ctx ? B->ctx->execKernel : 0u;
    p.seqTablesFromDict = B->seqTablesFromDict ? 1u : 0u;
    p.litDictSingleStream = B->litDictSingleStream ? 1u : 0u;
    if (B->ddict) {
        p.dict = B->ddict->dDict;
        p.dictContent = B->ddict->dBytes + B->ddict->entropySize;
        p.dictContentSize = B->ddict->size - B->ddict->entropySize;
    }
#ifdef ZG_STAMPS
    static uint64_t* dbgBuf = nullptr; static size_t dbgCap = 0;
    if (dbgCap < p.nBlocks) { hipFree(dbgBuf); hipMalloc((void**)&dbgBuf, (size_t)p.nBlocks * 512); dbgCap = p.nBlocks; }
    hipMemsetAsync(dbgBuf, 0, (size_t)p.nBlocks * 512, s);
    p.dbg = dbgBuf;
#endif
    // the scratch of the batch is shared by its launches: one on another stream than the last waits for that one
    if (B->launched && s != B->lastStream) hipStreamWaitEvent(s, B->evDone, 0);
    bool const prof = B->profiling && B->evReady;
    if (prof) hipEventRecord(B->ev[0], s);
    static_assert(sizeof(DevBlockState) % 16 == 0, "block states are zeroed 16 bytes at a time");
    if (p.nBlocks) launch_zero(B->dState, (size_t)p.nBlocks * sizeof(DevBlockState), s);
    // (a kernel, not a memset: replayed from a captured graph, a memset node of a few bytes was seen not to take effect on
    // the second launch -- ROCm 7.2, gfx950; the list's first 16 bytes are its count and entries written before they are read)
    launch_zero(B->dDryList, 16, s);
    launch_literals(p, s);
    if (prof) hipEventRecord(B->ev[1], s);
    // the pack of split-decoded literals needs nothing from the sequence stage and leaves it most of the chip: side stream
    hipEventRecord(B->evFork, s);
    hipStreamWaitEvent(B->side, B->evFork, 0);
    launch_pack_literals(p, B->side);
    hipEventRecord(B->evJoin, B->side);
    launch_sequences(p, s);
    launch_sequences_dry(p, s);
    if (prof) hipEventRecord(B->ev[2], s);
    launch_layout(p, s);
    if (prof) hipEventRecord(B->ev[3], s);
    hipStreamWaitEvent(s, B->evJoin, 0);
    launch_execute(p, s);
    if (prof) hipEventRecord(B->ev[4], s);
    launch_checksum(p, s);
    if (prof) { hipEventRecord(B->ev[5], s); B->evRecorded = true; }
    hipEventRecord(B->evDone, s);
    B->lastStream = s; B->launched = true;
#ifdef ZG_STAMPS
    {   // diagnostic build: print where the sequence stage spends its cycles
        hipStreamSynchronize(s);
        std::vector<uint64_t> h((size_t)p.nBlocks * 64);
        hipMemcpy(h.data(), dbgBuf, h.size() * 8, hipMemcpyDeviceToHost);
        {   // literal stage, the blocks with the most literals (they set the kernel's duration)
            double u[12] = {0}; size_t m = 0; uint64_t big = 0;
            for (size_t i = 0; i < p.nBlocks; i++) { const uint64_t* q = h.data() + 16 * p.nBlocks + 16 * i; if (q[8] > big) big = q[8]; }
            for (size_t i = 0; i < p.nBlocks; i++) { const uint64_t* q = h.data() + 16 * p.nBlocks + 16 * i; if (q[7] && q[8] * 10 >= big * 9) { for (int j = 0; j < 12; j++) u[j] += q[j]; m++; } }
            if (m) fprintf(stderr, "[stamps] k_literals, %zu largest blocks (%.0f literal bytes): table %.0f  loads+flush %.0f  sync+window-write %.0f  decode %.0f  loop %.0f  total %.0f ticks; %.0f rounds, %.0f restaged; table = parse %.0f (weights %.0f) + fill %.0f\n",
                           m, u[8] / m, u[0] / m, u[1] / m, u[2] / m, u[3] / m, u[4] / m, u[7] / m, u[5] / m, u[6] / m, u[9] / m, u[11] / m, u[10] / m);
        }
        {
            size_t hist[128] = {0}; size_t shown = 0;
            for (size_t i = 0; i < p.nBlocks; i++) { const uint64_t* q = h.data() + 16 * p.nBlocks + 16 * i; hist[q[12] & 127]++;
                if ((q[12] & 7) && shown < 6) { fprintf(stderr, "[stamps] split decode block %zu: why 0x%llx sum %llu want %llu idx/valid %llx\n", i, (unsigned long long)q[12], (unsigned long long)q[13], (unsigned long long)q[14], (unsigned long long)q[15]); shown++; } }
            for (int w = 0; w < 128; w++) if (hist[w]) fprintf(stderr, "[stamps] split decode why 0x%x: %zu blocks\n", w, hist[w]);
        }
        {   // sequence stage: the blocks with the most sequences
            double u[8] = {0}; size_t m = 0; uint64_t big = 0;
            for (size_t i = 0; i < p.nBlocks; i++) { const uint64_t* q = h.data() + 8 * p.nBlocks + 8 * i; if (q[4] > big) big = q[4]; }
            for (size_t i = 0; i < p.nBlocks; i++) { const uint64_t* q = h.data() + 8 * p.nBlocks + 8 * i; if (q[4] && q[4] * 100 >= big * 99) { for (int j = 0; j < 8; j++) u[j] += q[j]; m++; } }
            if (m) fprintf(stderr, "[stamps] k_sequences, %zu longest blocks (%.0f sequences): setup %.0f (descriptions staged at %.0f, tables built at %.0f, of which descriptions parsed by one lane %.0f)  in chain loop %.0f  whole kernel %.0f ticks; %.0f rounds\n", m, u[4] / m, u[0] / m, u[5] / m, u[6] / m, u[7] / m, u[1] / m, u[2] / m, u[3] / m);
        }
        if (p.execMode == 2) {
            double t[32] = {0}; size_t n = 0;
            for (size_t i = 0; i < p.nFrames; i++) if (h[32 * p.nBlocks + 32 * i + 6]) { for (int j = 0; j < 32; j++) t[j] += h[32 * p.nBlocks + 32 * i + j]; n++; }
            if (n) fprintf(stderr, "[stamps] k_execute_wf per frame (summed over its waves): plan + early part %.0f  wait for turn %.0f  late part %.0f  flush %.0f ticks; %.0f batches (%.0f the slow way), late ring matches %.0f, dependent matches %.0f, late memory matches %.0f; memory-to-memory batches %.0f (plan + wait %.0f, copy %.0f ticks); from one batch complete to the next %.0f ticks, of which until the next wave notices %.0f; slow way because: late memory match: %.1f, dependent match out of the ring's reach or over 64 bytes: %.1f, floor moved: %.1f; early part = records %.0f + plan %.0f + ring space, control words, load issue %.0f + dependent rounds %.0f + late bytes dealt out %.0f + stage and runs %.0f + long runs %.0f\n",
                           t[0] / n, t[1] / n, t[2] / n, t[3] / n, t[6] / n, t[5] / n, t[7] / n, t[8] / n, t[9] / n, t[10] / n, t[11] / n, t[12] / n, t[14] / n, t[15] / n, t[17] / n, t[18] / n, t[21] / n, t[22] / n, t[23] / n, t[24] / n, t[25] / n, t[26] / n, t[27] / n, t[28] / n);
        } else {
        double t[8] = {0}; size_t n = 0;
        for (size_t i = 0; i < p.nFrames; i++) if (h[8 * i + 5]) { for (int j = 0; j < 8; j++) t[j] += h[8 * i + j]; n++; }
        if (n) fprintf(stderr, "[stamps] k_execute per frame: setup %.0f  load issue + plan + wait + stage writes %.0f  runs %.0f  dependent matches %.0f  store %.0f ticks; %.0f batches, %.0f dependent matches, rest %.0f ticks\n",
                       t[0] / n, t[1] / n, t[2] / n, t[3] / n, t[4] / n, t[5] / n, t[6] / n, t[7] / n);
        }
    }
#endif
    return hipGetLastError() == hipSuccess ? 0 : mkerr(E_generic);
}

extern "C" size_t zstdgpu_batch_status(zstdgpu_batch* B, zstdgpu_frame_status* status)
{
    if (!B || !set_device(B)) return mkerr(E_generic);
    if (!wait_done(B)) return mkerr(E_generic);
    static_assert(sizeof(zstdgpu_frame_status) == sizeof(DevFrameStatus), "status layout");
    if (status && B->nFrames &&
        hipMemcpy(status, B->dStatus, (size_t)B->nFrames * sizeof(DevFrameStatus), hipMemcpyDeviceToHost) != hipSuccess)
        return mkerr(E_generic);
    return 0;
}

extern "C" uint32_t zstdgpu_batch_nFrames(const zstdgpu_batch* b) { return b->nFrames; }
extern "C" uint32_t zstdgpu_batch_nBlocks(const zstdgpu_batch* b) { return b->nBlocks; }
extern "C" uint64_t zstdgpu_batch_srcBytes(const zstdgpu_batch* b) { return b->srcBytes; }
extern "C" uint64_t zstdgpu_batch_scratchBytes(const zstdgpu_batch* b) { return b->scratchBytes; }

extern "C" void zstdgpu_batch_set_profiling(zstdgpu_batch* B, int enabled)
{
    if (!B || !set_device(B)) return;
    if (enabled && !B->evReady) {
        bool ok = true;
        for (auto& e : B->ev) ok = ok && hipEventCreate(&e) == hipSuccess;
        B->evReady = ok;
    }
    B->profiling = enabled != 0;
    B->evRecorded = false;
}
extern "C" size_t zstdgpu_batch_stage_ms(zstdgpu_batch* B, float* ms)
{
    if (!B || !B->evRecorded || !set_device(B)) return mkerr(E_generic);
    if (hipEventSynchronize(B->ev[5]) != hipSuccess) return mkerr(E_generic);
    for (int i = 0; i < 5; i++) if (hipEventElapsedTime(&ms[i], B->ev[i], B->ev[i + 1]) != hipSuccess) return mkerr(E_generic);
    if (hipEventElapsedTime(&ms[ZSTDGPU_STAGE_TOTAL], B->ev[0], B->ev[5]) != hipSuccess) return mkerr(E_generic);
    return 0;
}
extern "C" const char* zstdgpu_stage_kernel_name(int stage)
{
    static const char* names[] = {"k_literals", "k_sequences", "k_layout", "k_execute", "k_checksum", "total"};
    return stage >= 0 && stage < ZSTDGPU_NUM_STAGES ? names[stage] : "";
}

// ===================================================================== stage hooks
extern "C" size_t zstdgpu_xxh64Batch(zstdgpu_ctx* ctx, const void* devBase, const uint64_t* off, const uint64_t* len, uint32_t n, uint64_t* digests)
{
    if (!ctx || !set_device(ctx)) return mkerr(E_generic);
    if (!n) return 0;
    uint64_t *dOff = nullptr, *dLen = nullptr, *dOut = nullptr;
    uint64_t acct = 0;
    size_t rc = 0;
    if (!dalloc(dOff, n, acct) || !dalloc(dLen, n, acct) || !dalloc(dOut, n, acct)) rc = mkerr(E_memory_allocation);
    if (!rc && (hipMemcpy(dOff, off, n * 8, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(dLen, len, n * 8, hipMemcpyHostToDevice) != hipSuccess)) rc = mkerr(E_generic);
    if (!rc) {
        launch_xxh64((const uint8_t*)devBase, dOff, dLen, n, dOut, nullptr);
        if (hipMemcpy(digests, dOut, n * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = mkerr(E_generic);
    }
    hipFree(dOff); hipFree(dLen); hipFree(dOut);
    return rc;
}

extern "C" size_t zstdgpu_predefinedTables(zstdgpu_ctx* ctx, uint32_t* cells)
{
    if (!ctx || !cells || !set_device(ctx)) return mkerr(E_generic);
    uint32_t* d = nullptr;
    uint64_t acct = 0;
    if (!dalloc(d, 3 * 64 * 4, acct)) return mkerr(E_memory_allocation);
    launch_basic_tables(d, nullptr);
    size_t const rc = hipMemcpy(cells, d, 3 * 64 * 4 * sizeof(uint32_t), hipMemcpyDeviceToHost) == hipSuccess ? 0 : mkerr(E_generic);
    hipFree(d);
    return rc;
}

extern "C" size_t zstdgpu_batch_blockInfo(zstdgpu_batch* B, uint32_t bi, uint32_t* litSize, uint32_t* nbSeq, uint64_t* regenSize, uint32_t* errorCode)
{
    if (!B || bi >= B->nBlocks || !set_device(B) || !mirror(B)) return mkerr(E_generic);
    if (!wait_done(B)) return mkerr(E_generic);
    DevBlockState st;
    if (hipMemcpy(&st, B->dState + bi, sizeof(st), hipMemcpyDeviceToHost) != hipSuccess) return mkerr(E_generic);
    const DevBlock& d = B->hBlocks[bi];
    if (litSize) *litSize = d.litSize;
    if (nbSeq) *nbSeq = d.nbSeq;
    if (regenSize) *regenSize = st.regen;
    if (errorCode) *errorCode = d.hostLitErr ? d.hostLitErr : st.litErr ? st.litErr : d.hostSeqErr ? d.hostSeqErr : st.seqErr ? st.seqErr : st.tailErr;
    return 0;
}

extern "C" size_t zstdgpu_batch_copyLiterals(zstdgpu_batch* B, const void* devSrc, uint32_t bi, void* out, size_t cap)
{
    if (!B || bi >= B->nBlocks || !set_device(B) || !mirror(B)) return mkerr(E_generic);
    if (!wait_done(B)) return mkerr(E_generic);
    const DevBlock& d = B->hBlocks[bi];
    if (d.blockType != BT_COMPRESSED || d.hostLitErr) return mkerr(E_generic);
    if (d.litSize > cap) return mkerr(E_dstSize_tooSmall);
    if (!d.litSize) return 0;
    if (d.litType >= SET_COMPRESSED) {
        if (hipMemcpy(out, B->dLitBuf + d.litBufOff, d.litSize, hipMemcpyDeviceToHost) != hipSuccess) return mkerr(E_generic);
    } else if (d.litType == SET_BASIC) {
        if (hipMemcpy(out, (const uint8_t*)devSrc + d.srcOff + d.litHdrSize, d.litSize, hipMemcpyDeviceToHost) != hipSuccess) return mkerr(E_generic);
    } else {
        uint8_t v;
        if (hipMemcpy(&v, (const uint8_t*)devSrc + d.srcOff + d.litHdrSize, 1, hipMemcpyDeviceToHost) != hipSuccess) return mkerr(E_generic);
        memset(out, v, d.litSize);
    }
    return d.litSize;
}

extern "C" size_t zstdgpu_batch_copySequences(zstdgpu_batch* B, uint32_t bi, uint32_t* triples, size_t tripleCapacity)
try {
    if (!B || bi >= B->nBlocks || !set_device(B) || !mirror(B)) return mkerr(E_generic);
    if (!wait_done(B)) return mkerr(E_generic);
    const DevBlock& d = B->hBlocks[bi];
    if (d.blockType != BT_COMPRESSED || d.hostLitErr || d.hostSeqErr) return mkerr(E_generic);
    if (d.nbSeq > tripleCapacity) return mkerr(E_dstSize_tooSmall);
    if (!d.nbSeq) return 0;
    DevBlockState st;
    if (hipMemcpy(&st, B->dState + bi, sizeof(st), hipMemcpyDeviceToHost) != hipSuccess) return mkerr(E_generic);
    std::vector<uint32_t> rec(((size_t)d.nbSeq + 1) * 4);
    if (hipMemcpy(rec.data(), B->dSeqBuf + d.seqBufOff * 4, rec.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return mkerr(E_generic);
    // device records are (offset, output position, literal position); lengths are differences of neighbours
    for (uint32_t i = 0; i < st.seqValid; i++) {
        uint32_t x = rec[4 * i];
        if (x & kSymbolicRep) {   // offsets that referred to the incoming rep stack become concrete here
            uint32_t const v = st.repIn[(x >> 29) & 3], dl = x & ((1u << 29) - 1);
            x = v > dl ? v - dl : 1;
        }
        uint32_t const ll = rec[4 * i + 3];
        triples[3 * i] = ll;
        triples[3 * i + 1] = rec[4 * (i + 1) + 1] - rec[4 * i + 1] - ll;
        triples[3 * i + 2] = x;
    }
    return st.seqValid;
}
catch (...) { return mkerr(E_memory_allocation); }      // the C ABI never throws (std::bad_alloc from the host containers)

// ===================================================================== one-shot drop-ins
namespace {

bool ensure(uint8_t*& p, size_t& cap, size_t need)
{
    if (need <= cap && p) return true;
    hipFree(p); p = nullptr; cap = 0;
    size_t const want = need + need / 4 + 4096;
    if (hipMalloc((void**)&p, want) != hipSuccess) { p = nullptr; return false; }
    cap = want;
    return true;
}

// One group of frames of a one-shot call, source already in ctx->dSrc, output into ctx->dDst.  Small inputs (the limits of
// the no-allocation entry point) go through the device walk inside a fixed scratch area that the context allocates once:
// no batch object, no allocation, no stream or event is made per call -- what a caller decoding one small frame per call
// pays for otherwise (the reference's own per-call cost is the allocation of its context, zstd_decompressf.pas:977-988).
// An input that does not fit that area is known on the device only (every frame comes back memory_allocation): it is
// decoded again as a batch.  The two walks build the same descriptors, so the verdicts are the same either way.
size_t decode_group(zstdgpu_ctx* ctx, const uint8_t* src, size_t srcSize, const zstdgpu_frame_in* in, uint32_t n, const zstdgpu_ddict* ddict,
                    zstdgpu_frame_status* st)
{
    if (ctx->oneShotFixed && n <= ZSTDGPU_NOALLOC_MAX_FRAMES && srcSize <= ZSTDGPU_NOALLOC_MAX_SRC_BYTES) {
        if (!ctx->oneShotTried) {
            ctx->oneShotTried = true;
            if (hipMalloc((void**)&ctx->oneShotScratch, fixed_scratch_bytes()) != hipSuccess) ctx->oneShotScratch = nullptr;
            if (hipMalloc((void**)&ctx->oneShotIO, ZSTDGPU_NOALLOC_MAX_FRAMES * (sizeof(zstdgpu_frame_in) + sizeof(zstdgpu_frame_status))) != hipSuccess) ctx->oneShotIO = nullptr;
            (void)hipGetLastError();
        }
        if (ctx->oneShotScratch && ctx->oneShotIO) {
            zstdgpu_frame_in* const dIn = (zstdgpu_frame_in*)ctx->oneShotIO;
            zstdgpu_frame_status* const dSt = (zstdgpu_frame_status*)(ctx->oneShotIO + ZSTDGPU_NOALLOC_MAX_FRAMES * sizeof(zstdgpu_frame_in));
            if (hipMemcpy(dIn, in, (size_t)n * sizeof(zstdgpu_frame_in), hipMemcpyHostToDevice) != hipSuccess) return mkerr(E_generic);
            size_t const rc = decode_frames_fixed(ctx, ctx->oneShotScratch, ctx->dSrc, srcSize, dIn, n, ctx->dDst, dSt, ddict, nullptr);
            if (rc) return rc;
            if (hipMemcpy(st, dSt, (size_t)n * sizeof(zstdgpu_frame_status), hipMemcpyDeviceToHost) != hipSuccess) return mkerr(E_generic);
            bool refused = false;      // (no other path to this code: a frame with a header error keeps that error, the others are refused)
            for (uint32_t k = 0; k < n; k++) refused = refused || st[k].errorCode == E_memory_allocation;
            if (!refused) return 0;
        }
    }
    zstdgpu_batch* B = zstdgpu_batch_create_from_frames(ctx, src, srcSize, in, n, ddict);
    if (!B) return mkerr(E_memory_allocation);
    size_t rc = zstdgpu_batch_decode(B, ctx->dSrc, ctx->dDst, nullptr);
    if (!rc) rc = zstdgpu_batch_status(B, st);
    zstdgpu_batch_destroy(B);
    return rc;
}

// ZSTD_decompressMultiFrame (zstd_decompressf.pas:836-942) over the batched device path.
size_t decompress_multi(zstdgpu_ctx* ctx, void* dst, size_t dstCapacity, const void* srcv, size_t srcSize, const zstdgpu_ddict* ddict)
try {
    if (!ctx || !set_device(ctx)) return mkerr(E_generic);
    if (ddict && ddict->status) return mkerr(ddict->status);
    const uint8_t* const src = (const uint8_t*)srcv;
    if (srcSize > 0x7FFFFFFFu || dstCapacity > 0x7FFFFFFEu) { if (srcSize > 0x7FFFFFFFu) return mkerr(E_srcSize_wrong); dstCapacity = 0x7FFFFFFEu; }
    // split into frames on the host
    std::vector<zstdgpu_frame_in> ins;
    std::vector<uint64_t> known;      // content size of each frame or UNKNOWN
    size_t ip = 0;
    size_t tailErr = 0;               // error to report once the frames before it decoded fine
    bool more1 = false;
    bool const magicless = ctx->format != 0;
    while (srcSize - ip >= (magicless ? 1u : 5u)) {      // ZSTD_startingInputLength(format) (:857)
        uint32_t const magic = srcSize - ip >= 4 ? rd32(src + ip) : 0;
        if ((magic & kMagicSkipMask) == kMagicSkipStart) {
            size_t const sk = skippable_size(src + ip, srcSize - ip);
            if (iserr(sk)) { tailErr = sk; break; }
            ip += sk;
            continue;
        }
        size_t const fsz = find_frame_size(src + ip, srcSize - ip, magicless);
        if (iserr(fsz)) {
            // let the walk of this (last) frame produce the precise error in decode order
            uint32_t const code = zstdgpu_getErrorCode(fsz);
            if (code == E_prefix_unknown && more1) { tailErr = mkerr(E_srcSize_wrong); break; }   // :917-918
            ins.push_back({ip, srcSize - ip, 0, 0});
            known.push_back(frame_content_size(src + ip, srcSize - ip, magicless));
            ip = srcSize;
            break;
        }
        ins.push_back({ip, fsz, 0, 0});
        known.push_back(frame_content_size(src + ip, fsz, magicless));
        ip += fsz;
        more1 = true;
    }
    if (!tailErr && ip != srcSize) tailErr = mkerr(E_srcSize_wrong);   // :938-939
    if (ins.empty()) return tailErr ? tailErr : 0;

    if (!ensure(ctx->dSrc, ctx->dSrcCap, srcSize + 16) || !ensure(ctx->dDst, ctx->dDstCap, dstCapacity + 16)) return mkerr(E_memory_allocation);
    if (hipMemcpy(ctx->dSrc, src, srcSize, hipMemcpyHostToDevice) != hipSuccess) return mkerr(E_generic);

    // Frames whose content size is in the header can be placed up front and decoded together; a frame
    // of unknown size ends a group because the next frame's offset depends on its result.
    size_t produced = 0, i = 0;
    while (i < ins.size()) {
        size_t j = i;
        size_t off = produced;
        bool overflow = false;
        while (j < ins.size()) {
            ins[j].dstOff = off; ins[j].dstCapacity = dstCapacity - off;
            uint64_t const k = known[j];
            j++;
            if (k == ZSTDGPU_CONTENTSIZE_UNKNOWN || k == ZSTDGPU_CONTENTSIZE_ERROR) break;
            if (k > dstCapacity - off) { overflow = true; break; }   // this frame will report dstSize_tooSmall; later ones are never reached
            off += (size_t)k;
        }
        (void)overflow;
        uint32_t const n = (uint32_t)(j - i);
        std::vector<zstdgpu_frame_status> st(n);
        size_t const rc = decode_group(ctx, src, srcSize, ins.data() + i, n, ddict, st.data());
        if (rc) return rc;
        for (uint32_t k = 0; k < n; k++) {
            if (st[k].errorCode) {
                // as in the reference, the frames decoded before the failing one have reached dst
                if (produced) hipMemcpy(dst, ctx->dDst, produced, hipMemcpyDeviceToHost);
                if (st[k].errorCode == E_prefix_unknown && (i + k) > 0) return mkerr(E_srcSize_wrong);
                return mkerr(st[k].errorCode);
            }
            produced += (size_t)st[k].regenSize;
        }
        i = j;
    }
    if (produced && hipMemcpy(dst, ctx->dDst, produced, hipMemcpyDeviceToHost) != hipSuccess) return mkerr(E_generic);
    if (tailErr) return tailErr;
    return produced;
}
catch (...) { return mkerr(E_memory_allocation); }      // the C ABI never throws (std::bad_alloc from the host containers)

}  // namespace

extern "C" size_t zstdgpu_decompressDCtx(zstdgpu_ctx* ctx, void* dst, size_t dstCapacity, const void* src, size_t srcSize)
{
    return decompress_multi(ctx, dst, dstCapacity, src, srcSize, ctx ? ctx->refDDict : nullptr);   // ZSTD_getDDict (:951-969)
}
extern "C" size_t zstdgpu_decompress_usingDDict(zstdgpu_ctx* ctx, void* dst, size_t dstCapacity, const void* src, size_t srcSize, const zstdgpu_ddict* ddict)
{
    return decompress_multi(ctx, dst, dstCapacity, src, srcSize, ddict);
}
extern "C" size_t zstdgpu_decompress_usingDict(zstdgpu_ctx* ctx, void* dst, size_t dstCapacity, const void* src, size_t srcSize, const void* dict, size_t dictSize)
{
    if (!dict || !dictSize) return decompress_multi(ctx, dst, dstCapacity, src, srcSize, nullptr);
    zstdgpu_ddict* d = zstdgpu_createDDict(ctx, dict, dictSize);
    if (!d) return mkerr(E_memory_allocation);
    size_t const r = decompress_multi(ctx, dst, dstCapacity, src, srcSize, d);
    zstdgpu_freeDDict(d);
    return r;
}

// ===================================================================== streaming over the batched path
// ZSTD_decompressStream (decompress/zstd_decompressf.pas:1978-2359) as a host state machine in front of the one-shot
// path: the reference decodes block by block as input arrives; the kernels here want whole frames, so the stream
// collects one frame (its size is found by following the block headers as they arrive, the loop of
// ZSTD_findFrameSizeInfo, :566-639), decodes it, and hands the output over as room is given.  Results are the
// reference's: the same bytes, 0 returned when a frame has been decoded and flushed completely, a positive hint while
// more input or more room is needed, the same error codes.
// In the reference the stream *is* the decoding context (ZSTD_DStream = ZSTD_DCtx): its dictionary and parameters are
// the context's, so they live in zstdgpu_ctx here too.
struct zstdgpu_dstream {
    zstdgpu_ctx* ctx = nullptr;
    std::vector<uint8_t> in, out;
    size_t want = 5;          // bytes of `in` needed before the next decision (ZSTD_startingInputLength to begin with)
    size_t cursor = 0;        // offset of the next block header in `in` (0: frame header not seen yet)
    size_t blocks = 0;        // blocks seen
    bool checksum = false, skippable = false, lastSeen = false, onePass = false;
    uint64_t contentSize = ZSTDGPU_CONTENTSIZE_UNKNOWN, windowSize = 0;
    size_t outPos = 0;
    bool flushing = false;
    size_t error = 0;         // sticky until the stream is reset
    zstdgpu_outBuffer expected{nullptr, 0, 0};   // ZSTD_d_stableOutBuffer: the output buffer as the last call left it
    bool busy() const { return !in.empty() || flushing; }      // streamStage <> zdss_init
    void reset() { in.clear(); out.clear(); want = (ctx && ctx->format) ? 1 : 5; cursor = 0; blocks = 0; checksum = skippable = lastSeen = onePass = false;
                   contentSize = ZSTDGPU_CONTENTSIZE_UNKNOWN; windowSize = 0; outPos = 0; flushing = false; error = 0; }
};

static void stream_detach(zstdgpu_dstream* z) { z->ctx = nullptr; z->error = mkerr(E_stage_wrong); }

extern "C" zstdgpu_dstream* zstdgpu_createDStream(zstdgpu_ctx* ctx)
{
    if (!ctx) return nullptr;
    zstdgpu_dstream* z = new (std::nothrow) zstdgpu_dstream;
    if (!z) return nullptr;
    z->ctx = ctx; z->reset();
    try { ctx->streams.push_back(z); } catch (...) { delete z; return nullptr; }
    return z;
}
extern "C" void zstdgpu_freeDStream(zstdgpu_dstream* z)
{
    if (!z) return;
    if (z->ctx) { auto& v = z->ctx->streams; v.erase(std::remove(v.begin(), v.end(), z), v.end()); }
    delete z;
}
// ZSTD_initDStream / ZSTD_initDStream_usingDDict (:1605-1628): start a new stream (session reset + ZSTD_DCtx_refDDict);
// returns the first input size hint (ZSTD_startingInputLength of the context's format)
// The dictionary belongs to the context, which several streams may share here (in the reference a stream is the
// context): while another stream of the context is inside a frame the dictionary cannot change (stage_wrong).
static size_t init_stream(zstdgpu_dstream* z, const zstdgpu_ddict* ddict)
{
    if (!z || !z->ctx) return mkerr(z ? E_stage_wrong : E_generic);
    for (auto* o : z->ctx->streams) if (o != z && o->busy() && z->ctx->refDDict != ddict) return mkerr(E_stage_wrong);
    z->reset(); z->ctx->refDDict = ddict;
    return z->want;
}
extern "C" size_t zstdgpu_initDStream(zstdgpu_dstream* z) { return init_stream(z, nullptr); }
extern "C" size_t zstdgpu_initDStream_usingDDict(zstdgpu_dstream* z, const zstdgpu_ddict* ddict) { return init_stream(z, ddict); }
// ZSTD_resetDStream (:1630-1637): forget a half-read frame, keep the dictionary
extern "C" size_t zstdgpu_resetDStream(zstdgpu_dstream* z) { if (!z || !z->ctx) return mkerr(z ? E_stage_wrong : E_generic); z->reset(); return z->want; }

// ---- decoder parameters (zstd_decompressf.pas:1641-1826)
extern "C" zstdgpu_bounds zstdgpu_dParam_getBounds(zstdgpu_dParameter p)       // ZSTD_dParam_getBounds :1680-1715
{
    zstdgpu_bounds b{0, 0, 0};
    switch ((int)p) {
        case ZSTDGPU_d_windowLogMax: b.lowerBound = 10; b.upperBound = (int)kWindowLogMax; return b;   // ZSTD_WINDOWLOG_ABSOLUTEMIN .. ZSTD_WINDOWLOG_MAX
        case ZSTDGPU_d_format: b.lowerBound = 0; b.upperBound = 1; return b;
        case ZSTDGPU_d_stableOutBuffer: b.lowerBound = 0; b.upperBound = 1; return b;
        case ZSTDGPU_d_forceIgnoreChecksum: b.lowerBound = 0; b.upperBound = 1; return b;
        default: break;
    }
    b.error = mkerr(E_parameter_unsupported);
    return b;
}
namespace {
bool any_stream_busy(const zstdgpu_ctx* ctx) { for (auto* z : ctx->streams) if (z->busy()) return true; return false; }
}  // namespace
extern "C" size_t zstdgpu_DCtx_setParameter(zstdgpu_ctx* ctx, zstdgpu_dParameter p, int value)   // ZSTD_DCtx_setParameter :1763-1807
{
    if (!ctx) return mkerr(E_generic);
    if (any_stream_busy(ctx)) return mkerr(E_stage_wrong);
    zstdgpu_bounds const b = zstdgpu_dParam_getBounds(p);
    if (iserr(b.error)) return b.error;
    if ((int)p == ZSTDGPU_d_windowLogMax && value == 0) value = 27;     // ZSTD_WINDOWLOG_LIMIT_DEFAULT
    if (value < b.lowerBound || value > b.upperBound) return mkerr(E_parameter_outOfBound);
    switch ((int)p) {
        case ZSTDGPU_d_windowLogMax: ctx->maxWindowSize = 1ull << value; break;
        case ZSTDGPU_d_format: ctx->format = (uint32_t)value; for (auto* z : ctx->streams) z->reset(); break;
        case ZSTDGPU_d_stableOutBuffer: ctx->stableOut = (uint32_t)value; break;
        default: ctx->ignoreChecksum = (uint32_t)value; break;
    }
    return 0;
}
extern "C" size_t zstdgpu_DCtx_getParameter(zstdgpu_ctx* ctx, zstdgpu_dParameter p, int* value)   // ZSTD_DCtx_getParameter :1735-1761
{
    if (!ctx || !value) return mkerr(E_generic);
    switch ((int)p) {
        case ZSTDGPU_d_windowLogMax: { int l = 0; while ((ctx->maxWindowSize >> (l + 1)) != 0) l++; *value = l; return 0; }   // ZSTD_highbit32
        case ZSTDGPU_d_format: *value = (int)ctx->format; return 0;
        case ZSTDGPU_d_stableOutBuffer: *value = (int)ctx->stableOut; return 0;
        case ZSTDGPU_d_forceIgnoreChecksum: *value = (int)ctx->ignoreChecksum; return 0;
        default: return mkerr(E_parameter_unsupported);
    }
}
extern "C" size_t zstdgpu_DCtx_refDDict(zstdgpu_ctx* ctx, const zstdgpu_ddict* ddict)            // ZSTD_DCtx_refDDict :1641-1652
{
    if (!ctx) return mkerr(E_generic);
    if (any_stream_busy(ctx)) return mkerr(E_stage_wrong);
    ctx->refDDict = ddict;
    return 0;
}
// which execute kernel takes the frames of later launches (tests and measurements; not a reference function)
extern "C" size_t zstdgpu_test_setExecuteKernel(zstdgpu_ctx* ctx, int kernel)
{
    if (!ctx) return mkerr(E_generic);
    if (kernel < 0 || kernel > 2) return mkerr(E_parameter_outOfBound);
    ctx->execKernel = (uint32_t)kernel;
    return 0;
}
extern "C" size_t zstdgpu_DCtx_reset(zstdgpu_ctx* ctx, zstdgpu_ResetDirective reset)            // ZSTD_DCtx_reset :1809-1826
{
    if (!ctx) return mkerr(E_generic);
    if (reset == ZSTDGPU_reset_session_only || reset == ZSTDGPU_reset_session_and_parameters)
        for (auto* z : ctx->streams) z->reset();
    if (reset == ZSTDGPU_reset_parameters || reset == ZSTDGPU_reset_session_and_parameters) {
        if (any_stream_busy(ctx)) return mkerr(E_stage_wrong);
        ctx->refDDict = nullptr;
        ctx->reset_parameters();
        for (auto* z : ctx->streams) z->reset();
    }
    return 0;
}

extern "C" size_t zstdgpu_decompressStream(zstdgpu_dstream* z, zstdgpu_outBuffer* output, zstdgpu_inBuffer* input)
try {
    if (!z || !output || !input) return mkerr(E_generic);
    if (input->pos > input->size) return mkerr(E_srcSize_wrong);       // :1994-2003
    if (output->pos > output->size) return mkerr(E_dstSize_tooSmall);
    if (z->error) return z->error;
    if (!z->ctx) return mkerr(E_stage_wrong);
    auto fail = [&](uint32_t code) { z->error = mkerr(code); return z->error; };
    bool const stable = z->ctx->stableOut != 0, magicless = z->ctx->format != 0;
    // ZSTD_checkOutBuffer (:1913-1928): in stable mode the caller must hand the same buffer back once a frame has begun
    if (stable && z->busy() && (z->expected.dst != output->dst || z->expected.size != output->size || z->expected.pos != output->pos))
        return fail(E_dstBuffer_wrong);
    struct Remember { zstdgpu_dstream* z; zstdgpu_outBuffer* o; ~Remember() { z->expected = *o; } } remember{z, output};
    bool const fresh = !z->busy();            // this call starts at a frame boundary
    size_t const startPos = input->pos;
    for (;;) {
        if (z->flushing) {
            // stable mode decodes straight into the caller's buffer, which must hold what is left of the frame (:1941-1951)
            if (stable && z->out.size() - z->outPos > output->size - output->pos) return fail(E_dstSize_tooSmall);
            size_t const n = std::min(z->out.size() - z->outPos, output->size - output->pos);
            if (n) memcpy((uint8_t*)output->dst + output->pos, z->out.data() + z->outPos, n);
            output->pos += n; z->outPos += n;
            if (z->outPos < z->out.size()) return z->out.size() - z->outPos;      // more room needed
            z->reset();                                                           // (keeps the dictionary)
            return 0;                                                             // frame done
        }
        // collect what the next decision needs
        if (z->in.size() < z->want) {
            size_t const n = std::min(z->want - z->in.size(), input->size - input->pos);
            z->in.insert(z->in.end(), (const uint8_t*)input->src + input->pos, (const uint8_t*)input->src + input->pos + n);
            input->pos += n;
            if (z->in.size() < z->want) return z->want - z->in.size();            // more input needed
        }
        const uint8_t* const p = z->in.data();
        if (z->cursor == 0 && !z->skippable) {
            // frame start: magic, then the whole frame header
            if (!magicless && z->in.size() >= 4 && (rd32(p) & kMagicSkipMask) == kMagicSkipStart) {   // :301-318
                if (z->in.size() < 8) { z->want = 8; continue; }
                uint32_t const sz = rd32(p + 4);
                if ((uint32_t)(sz + 8) < sz) return fail(E_frameParameter_unsupported);
                z->skippable = true; z->want = (size_t)8 + sz;
                continue;
            }
            if (!magicless && z->in.size() >= 4 && rd32(p) != kMagic) return fail(E_prefix_unknown);
            size_t const fhsize = frame_header_size(p, z->in.size(), magicless);
            if (z->in.size() < fhsize) { z->want = fhsize; continue; }
            FrameHeader fh;
            size_t const r = get_frame_header(fh, p, z->in.size(), magicless);
            if (iserr(r)) return fail((uint32_t)(0 - r));
            // The reference decodes a frame in one pass, without its streaming window buffer and so without the window
            // limit, when the frame is whole in this call's input, its content size is known and the output has room
            // for it (:2111-2129); otherwise ZSTD_d_windowLogMax applies (:2166).
            bool const onePass = fresh && fh.contentSize != ZSTDGPU_CONTENTSIZE_UNKNOWN && output->size - output->pos >= fh.contentSize &&
                                 !iserr(find_frame_size((const uint8_t*)input->src + startPos, input->size - startPos, false));
            if (!onePass && fh.windowSize > z->ctx->maxWindowSize) return fail(E_frameParameter_windowTooLarge);
            z->checksum = fh.checksumFlag != 0; z->contentSize = fh.contentSize; z->windowSize = fh.windowSize; z->onePass = onePass;
            z->cursor = fhsize; z->want = fhsize + 3;
            continue;
        }
        if (z->skippable) {      // a skippable frame has arrived whole: it regenerates nothing (:885-893)
            z->flushing = true; z->out.clear(); z->outPos = 0;
            continue;
        }
        if (!z->lastSeen) {
            // the block header at `cursor` is here: it tells where the next one is (ZSTD_getcBlockSize, zstd_decompress_block.pas:190-207)
            uint32_t const hdr = rd24(p + z->cursor);
            uint32_t const type = (hdr >> 1) & 3, size = hdr >> 3;
            if (type == 3) return fail(E_corruption);
            size_t const cSize = type == BT_RLE ? 1 : size;
            z->blocks++;
            z->cursor += 3 + cSize;
            if (hdr & 1) { z->lastSeen = true; z->want = z->cursor + (z->checksum ? 4 : 0); }
            else z->want = z->cursor + 3;
            continue;
        }
        // the frame is complete: decode it
        {
            // Room for the output: the content size of the header when the caller has itself offered that much (one-pass
            // case above), else no more than the blocks that arrived can regenerate in a stream -- blockSizeMax each
            // (:1170); a header that promises more is wrong, and must not make this stream allocate what it names.
            uint64_t const byBlocks = (uint64_t)z->blocks * (z->windowSize < kBlockSizeMax ? z->windowSize : kBlockSizeMax);
            uint64_t bound = z->contentSize != ZSTDGPU_CONTENTSIZE_UNKNOWN ? z->contentSize : byBlocks;
            bool const capped = !z->onePass && bound > byBlocks;
            if (capped) bound = byBlocks;
            if (bound > 0x7FFFFFFEull) return fail(E_memory_allocation);
            z->out.resize((size_t)bound);
            size_t r = decompress_multi(z->ctx, z->out.data(), z->out.size(), p, z->in.size(), z->ctx->refDDict);
            if (capped && iserr(r) && zstdgpu_getErrorCode(r) == E_dstSize_tooSmall) r = mkerr(E_corruption);     // some block exceeds blockSizeMax (:1170)
            if (iserr(r)) { z->error = r; return r; }
            z->out.resize(r);
            z->outPos = 0; z->flushing = true;
        }
    }
}
catch (...) { return mkerr(E_memory_allocation); }      // the C ABI never throws (std::bad_alloc from the host containers)

// ===================================================================== bufferless, block at a time
// ZSTD_decompressBegin[_usingDict|_usingDDict] / ZSTD_nextSrcSizeToDecompress / ZSTD_nextInputType /
// ZSTD_decompressContinue (zstd_decompressf.pas:995-1246, :1386-1452): the caller hands over exactly the bytes asked
// for -- a frame header, a block header, a block, a checksum -- and gets every block's bytes back from the call that
// brought the block.  The stage machine is the reference's, on the host.  A compressed block is decoded on the GPU as
// the only block of a frame of its own whose "dictionary" is what the blocks before it left behind: the entropy state
// (Huffman table, three sequence tables, repeat offsets, in the cell formats of a digested dictionary; k_carry_entropy
// moves a block's own tables into it) and the frame's output so far as the dictionary's content (the window).  The
// kernels are the batch's; nothing in them knows about this interface.
// Where the reference keeps the window in the caller's buffers (ZSTD_checkContinuity, zstd_decompress_block.pas:1545-1554),
// the window here is a device buffer of the context: the last windowSize + dictionary bytes of output stay in it, and
// where the caller puts the pieces does not matter.
struct zstdgpu_bufferless {
    enum Stage { getFrameHeaderSize = 0, decodeFrameHeader, decodeBlockHeader, decompressBlock, decompressLastBlock, checkChecksum,
                 decodeSkippableHeader, skipFrame };       // ZSTD_dStage (zstd_ddict.pas:96-100)
    bool begun = false;
    Stage stage = getFrameHeaderSize;
    size_t expected = 0, headerSize = 0, rleSize = 0;
    uint8_t headerBuffer[18] = {0};                        // ZSTD_FRAMEHEADERSIZE_MAX
    FrameHeader fh;
    uint32_t blockSizeMax = 0, bType = 3, dictID = 0;
    bool validateChecksum = false;
    uint64_t decodedSize = 0;
    // the state a block hands to the next one, shaped as a dictionary for the kernels: dDict = entropy tables and repeat
    // offsets, dBytes[0, size) = the window (dictionary content, then the frame's output so far)
    zstdgpu_ddict state;
    size_t histCap = 0;
    size_t keep = 0;              // bytes before the next block that a match may reach: window size + dictionary content
    size_t dictContent = 0;
    DevXxhState* dXxh = nullptr;
    std::vector<uint8_t> mini;    // the block wrapped as a frame
};

static void bufferless_free(zstdgpu_bufferless* b)
{
    if (!b) return;
    hipFree(b->state.dBytes); hipFree(b->state.dDict); hipFree(b->dXxh);
    delete b;
}

namespace {

constexpr size_t kBlHeadBytes = offsetof(DevDict, huf);       // the scalar fields of the carried state

// room for n more bytes of window: grow the buffer until it holds twice what must be kept (plus two blocks), then move
// the part to keep to the front when it is full (source and destination of that copy cannot overlap then)
bool window_room(zstdgpu_bufferless* b, size_t n)
{
    size_t const len = b->state.size;
    if (len + n <= b->histCap) return true;
    size_t const K = b->keep;
    if (len > K && len - K >= K) {
        if (K && hipMemcpy(b->state.dBytes, b->state.dBytes + (len - K), K, hipMemcpyDeviceToDevice) != hipSuccess) return false;
        b->state.size = (uint32_t)K;
        if (K + n <= b->histCap) return true;
    }
    size_t const len2 = b->state.size;
    size_t const full = 2 * K + 2 * (size_t)kBlockSizeMax + 64;
    size_t want = std::max(len2 + n + 64, std::min(full, 2 * b->histCap + 65536));
    if (want > 0xFFFFFF00ull) return false;
    uint8_t* p = nullptr;
    if (hipMalloc((void**)&p, want) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (len2 && hipMemcpy(p, b->state.dBytes, len2, hipMemcpyDeviceToDevice) != hipSuccess) { hipFree(p); return false; }
    hipFree(b->state.dBytes);
    b->state.dBytes = p; b->histCap = want;
    return true;
}

// what ZSTD_decompressBegin leaves (zstd_decompressf.pas:1386-1408) plus, when given, the dictionary (:1358-1384, :1426-1452)
size_t bufferless_begin(zstdgpu_ctx* ctx, const zstdgpu_ddict* ddict)
try {
    if (!ctx || !set_device(ctx)) return mkerr(E_generic);
    if (ddict && ddict->status) return mkerr(E_dictionary_corrupted);
    zstdgpu_bufferless* b = ctx->bufferless;
    if (!b) {
        b = new (std::nothrow) zstdgpu_bufferless;
        if (!b) return mkerr(E_memory_allocation);
        b->state.ctx = nullptr; b->state.device = ctx->device;
        if (hipMalloc((void**)&b->state.dDict, sizeof(DevDict)) != hipSuccess || hipMalloc((void**)&b->dXxh, sizeof(DevXxhState)) != hipSuccess) {
            (void)hipGetLastError(); bufferless_free(b); return mkerr(E_memory_allocation);
        }
        ctx->bufferless = b;
    }
    b->begun = false;
    b->expected = ctx->format ? 1 : 5;          // ZSTD_startingInputLength (:96-107)
    b->stage = zstdgpu_bufferless::getFrameHeaderSize;
    b->decodedSize = 0; b->bType = 3; b->dictID = 0; b->validateChecksum = false;
    b->state.size = 0; b->state.entropySize = 0; b->state.dictID = 0; b->state.status = 0; b->state.hasEntropy = 1; b->state.entropyMask = 0;
    b->dictContent = 0; b->keep = 0;
    // the scalar fields of the carried state: repeat offsets {1, 4, 8} (zstd_internal.pas:13), no table set yet
    DevDict* const head = new (std::nothrow) DevDict;
    if (!head) return mkerr(E_memory_allocation);
    memset(head, 0, kBlHeadBytes);
    head->hasEntropy = 1; head->rep[0] = 1; head->rep[1] = 4; head->rep[2] = 8;
    bool ok = true;
    if (ddict && ddict->size) {
        if (ddict->hasEntropy) {
            // the dictionary's digested tables and repeat offsets become the state; its content becomes the window
            ok = hipMemcpy(b->state.dDict, ddict->dDict, sizeof(DevDict), hipMemcpyDeviceToDevice) == hipSuccess;
            b->state.entropyMask = 3;
            b->dictID = ddict->dictID;
        } else ok = hipMemcpy(b->state.dDict, head, kBlHeadBytes, hipMemcpyHostToDevice) == hipSuccess;
        size_t const content = ddict->size - ddict->entropySize;
        b->dictContent = content; b->keep = content;
        ok = ok && window_room(b, content);
        if (ok && content) ok = hipMemcpy(b->state.dBytes, ddict->dBytes + ddict->entropySize, content, hipMemcpyDeviceToDevice) == hipSuccess;
        if (ok) b->state.size = (uint32_t)content;
    } else ok = hipMemcpy(b->state.dDict, head, kBlHeadBytes, hipMemcpyHostToDevice) == hipSuccess;
    delete head;
    if (!ok) return mkerr(E_memory_allocation);
    b->begun = true;
    return 0;
}
catch (...) { return mkerr(E_memory_allocation); }

// ZSTD_nextSrcSizeToDecompressWithInputSize (:1010-1017): a raw block may arrive in pieces
size_t next_size_with_input(const zstdgpu_bufferless* b, size_t inputSize)
{
    if (!(b->stage == zstdgpu_bufferless::decompressBlock || b->stage == zstdgpu_bufferless::decompressLastBlock)) return b->expected;
    if (b->bType != BT_RAW) return b->expected;
    return std::min(std::max<size_t>(inputSize, 1), b->expected);
}

// window descriptor byte of the smallest window of at least w bytes (the block's own frame states the real frame's
// window, which decides whether zstd 1.4.8 may take its prefetching sequence loop: above 2^24)
uint8_t window_descriptor(uint64_t w)
{
    for (uint32_t e = 0; e <= 21; e++)
        for (uint32_t m = 0; m < 8; m++) {
            uint64_t const base = 1ull << (10 + e);
            if (base + (base >> 3) * m >= w) return (uint8_t)((e << 3) | m);
        }
    return (uint8_t)((21u << 3) | 7u);
}

// One compressed block on the GPU (ZSTD_decompressBlock_internal, zstd_decompress_block.pas:1491-1542, with the context's
// entropy state and window): returns the regenerated size or an error; the bytes are in ctx->dDst.
size_t bufferless_block(zstdgpu_ctx* ctx, zstdgpu_bufferless* b, size_t dstCapacity, const uint8_t* src, size_t srcSize)
{
    if (srcSize >= kBlockSizeMax) return mkerr(E_srcSize_wrong);        // :1513
    bool const magicless = ctx->format != 0;
    b->mini.clear();
    if (!magicless) { b->mini.push_back(0x28); b->mini.push_back(0xB5); b->mini.push_back(0x2F); b->mini.push_back(0xFD); }
    b->mini.push_back(0x00);                                           // descriptor: window byte follows, no content size, no checksum, no dictionary ID
    b->mini.push_back(window_descriptor(b->fh.windowSize));
    uint32_t const bh = 1u | ((uint32_t)BT_COMPRESSED << 1) | ((uint32_t)srcSize << 3);
    b->mini.push_back((uint8_t)bh); b->mini.push_back((uint8_t)(bh >> 8)); b->mini.push_back((uint8_t)(bh >> 16));
    b->mini.insert(b->mini.end(), src, src + srcSize);
    size_t const cap = std::min<size_t>(dstCapacity, 0x7FFFFFFEu);
    if (!ctx->oneShotTried) {
        ctx->oneShotTried = true;
        if (hipMalloc((void**)&ctx->oneShotScratch, fixed_scratch_bytes()) != hipSuccess) ctx->oneShotScratch = nullptr;
        if (hipMalloc((void**)&ctx->oneShotIO, ZSTDGPU_NOALLOC_MAX_FRAMES * (sizeof(zstdgpu_frame_in) + sizeof(zstdgpu_frame_status))) != hipSuccess) ctx->oneShotIO = nullptr;
        (void)hipGetLastError();
    }
    if (!ctx->oneShotScratch || !ctx->oneShotIO) return mkerr(E_memory_allocation);
    if (!ensure(ctx->dSrc, ctx->dSrcCap, b->mini.size() + 16) || !ensure(ctx->dDst, ctx->dDstCap, cap + 16)) return mkerr(E_memory_allocation);
    if (hipMemcpy(ctx->dSrc, b->mini.data(), b->mini.size(), hipMemcpyHostToDevice) != hipSuccess) return mkerr(E_generic);
    zstdgpu_frame_in const in{0, b->mini.size(), 0, cap};
    zstdgpu_frame_in* const dIn = (zstdgpu_frame_in*)ctx->oneShotIO;
    zstdgpu_frame_status* const dSt = (zstdgpu_frame_status*)(ctx->oneShotIO + ZSTDGPU_NOALLOC_MAX_FRAMES * sizeof(zstdgpu_frame_in));
    if (hipMemcpy(dIn, &in, sizeof(in), hipMemcpyHostToDevice) != hipSuccess) return mkerr(E_generic);
    size_t const rc = decode_frames_fixed(ctx, ctx->oneShotScratch, ctx->dSrc, b->mini.size(), dIn, 1, ctx->dDst, dSt, &b->state, nullptr, b->state.dDict);
    if (rc) return rc;
    zstdgpu_frame_status st{};
    if (hipMemcpy(&st, dSt, sizeof(st), hipMemcpyDeviceToHost) != hipSuccess) return mkerr(E_generic);
    if (st.errorCode) return mkerr(st.errorCode);
    // which tables the state holds now (the walk of the next block wants to know)
    uint32_t valid = 0;
    if (hipMemcpy(&valid, (const uint8_t*)b->state.dDict + offsetof(DevDict, entropyValid), 4, hipMemcpyDeviceToHost) != hipSuccess) return mkerr(E_generic);
    b->state.entropyMask = valid & 3u;
    return (size_t)st.regenSize;
}

}  // namespace

extern "C" size_t zstdgpu_decompressBegin(zstdgpu_ctx* ctx) { return bufferless_begin(ctx, nullptr); }
extern "C" size_t zstdgpu_decompressBegin_usingDDict(zstdgpu_ctx* ctx, const zstdgpu_ddict* ddict) { return bufferless_begin(ctx, ddict); }
extern "C" size_t zstdgpu_decompressBegin_usingDict(zstdgpu_ctx* ctx, const void* dict, size_t dictSize)
{
    if (!dict || !dictSize) return bufferless_begin(ctx, nullptr);
    zstdgpu_ddict* d = zstdgpu_createDDict(ctx, dict, dictSize);
    if (!d) return mkerr(E_memory_allocation);
    size_t const r = bufferless_begin(ctx, d);      // (the state holds copies: the dictionary object is not needed afterwards)
    zstdgpu_freeDDict(d);
    return r;
}
extern "C" size_t zstdgpu_nextSrcSizeToDecompress(zstdgpu_ctx* ctx)
{
    return (ctx && ctx->bufferless && ctx->bufferless->begun) ? ctx->bufferless->expected : 0;
}
extern "C" unsigned zstdgpu_nextInputType(zstdgpu_ctx* ctx)
{
    if (!ctx || !ctx->bufferless) return ZSTDGPU_nit_frameHeader;
    switch (ctx->bufferless->stage) {
        default:
        case zstdgpu_bufferless::getFrameHeaderSize: case zstdgpu_bufferless::decodeFrameHeader: return ZSTDGPU_nit_frameHeader;
        case zstdgpu_bufferless::decodeBlockHeader: return ZSTDGPU_nit_blockHeader;
        case zstdgpu_bufferless::decompressBlock: return ZSTDGPU_nit_block;
        case zstdgpu_bufferless::decompressLastBlock: return ZSTDGPU_nit_lastBlock;
        case zstdgpu_bufferless::checkChecksum: return ZSTDGPU_nit_checksum;
        case zstdgpu_bufferless::decodeSkippableHeader: case zstdgpu_bufferless::skipFrame: return ZSTDGPU_nit_skippableFrame;
    }
}

extern "C" size_t zstdgpu_decompressContinue(zstdgpu_ctx* ctx, void* dstv, size_t dstCapacity, const void* srcv, size_t srcSize)
try {
    typedef zstdgpu_bufferless BL;
    if (!ctx || !set_device(ctx)) return mkerr(E_generic);
    BL* const b = ctx->bufferless;
    if (!b || !b->begun) return mkerr(E_stage_wrong);
    uint8_t* const dst = (uint8_t*)dstv;
    const uint8_t* const src = (const uint8_t*)srcv;
    bool const magicless = ctx->format != 0;
    if (srcSize != next_size_with_input(b, srcSize)) return mkerr(E_srcSize_wrong);
    if (srcSize && !src) return mkerr(E_generic);
    switch (b->stage) {
    case BL::getFrameHeaderSize:
        if (!magicless) {
            if (srcSize < 4) return mkerr(E_srcSize_wrong);      // (the reference reads the magic number whatever it was given)
            if ((rd32(src) & kMagicSkipMask) == kMagicSkipStart) {
                memcpy(b->headerBuffer, src, srcSize);
                b->expected = 8 - srcSize;                       // ZSTD_SKIPPABLEHEADERSIZE
                b->stage = BL::decodeSkippableHeader;
                return 0;
            }
        }
        b->headerSize = frame_header_size(src, srcSize, magicless);
        if (iserr(b->headerSize)) return b->headerSize;
        memcpy(b->headerBuffer, src, srcSize);
        b->expected = b->headerSize - srcSize;
        b->stage = BL::decodeFrameHeader;
        return 0;
    case BL::decodeFrameHeader: {
        memcpy(b->headerBuffer + (b->headerSize - srcSize), src, srcSize);
        // ZSTD_decodeFrameHeader (:541-555; the dictionary ID check is zstd 1.4.8's, which the port dropped)
        b->fh = FrameHeader{};
        size_t const r = get_frame_header(b->fh, b->headerBuffer, b->headerSize, magicless);
        if (iserr(r)) return r;
        if (r > 0) return mkerr(E_srcSize_wrong);
        if (b->fh.dictID && b->dictID != b->fh.dictID) return mkerr(E_dictionary_wrong);
        b->validateChecksum = b->fh.checksumFlag && !ctx->ignoreChecksum;
        if (b->validateChecksum) {
            DevXxhState x{};                                     // XXH64_reset(seed 0) (xxhash.pas:533-545)
            x.v[0] = 0x9E3779B185EBCA87ull + 0xC2B2AE3D27D4EB4Full; x.v[1] = 0xC2B2AE3D27D4EB4Full; x.v[2] = 0; x.v[3] = 0ull - 0x9E3779B185EBCA87ull;
            x.digest = 0xEF46DB3751D8E999ull;                    // XXH64 of nothing
            if (hipMemcpy(b->dXxh, &x, sizeof(x), hipMemcpyHostToDevice) != hipSuccess) return mkerr(E_generic);
        }
        b->blockSizeMax = (uint32_t)std::min<uint64_t>(b->fh.windowSize, kBlockSizeMax);
        // (the window buffer's length is 32-bit, as the port's sizes are, and must hold what is kept twice over: no more
        // than 1.5 GB of output stays reachable, whatever window the header states; the buffer grows as output arrives)
        b->keep = (size_t)std::min<uint64_t>(b->fh.windowSize + b->dictContent, 0x60000000ull);
        b->expected = 3;
        b->stage = BL::decodeBlockHeader;
        return 0;
    }
    case BL::decodeBlockHeader: {
        // ZSTD_getcBlockSize (zstd_decompress_block.pas:190-207)
        uint32_t const hdr = rd24(src);
        uint32_t const last = hdr & 1, type = (hdr >> 1) & 3, size = hdr >> 3;
        if (type == 3) return mkerr(E_corruption);
        size_t const cBlockSize = type == BT_RLE ? 1 : size;
        if (cBlockSize > b->blockSizeMax) return mkerr(E_corruption);
        b->expected = cBlockSize; b->bType = type; b->rleSize = size;
        if (cBlockSize) { b->stage = last ? BL::decompressLastBlock : BL::decompressBlock; return 0; }
        if (last) {      // empty block
            if (b->fh.checksumFlag) { b->expected = 4; b->stage = BL::checkChecksum; }
            else { b->expected = 0; b->stage = BL::getFrameHeaderSize; }
        } else { b->expected = 3; b->stage = BL::decodeBlockHeader; }
        return 0;
    }
    case BL::decompressLastBlock:
    case BL::decompressBlock: {
        size_t rSize;
        if (b->bType == BT_COMPRESSED) {
            rSize = bufferless_block(ctx, b, dstCapacity, src, srcSize);
            if (iserr(rSize)) return rSize;
            b->expected = 0;
            if (rSize > b->blockSizeMax) return mkerr(E_corruption);
            if (!window_room(b, rSize)) return mkerr(E_memory_allocation);
            if (rSize && hipMemcpy(b->state.dBytes + b->state.size, ctx->dDst, rSize, hipMemcpyDeviceToDevice) != hipSuccess) return mkerr(E_generic);
            if (rSize && hipMemcpy(dst, ctx->dDst, rSize, hipMemcpyDeviceToHost) != hipSuccess) return mkerr(E_generic);
        } else if (b->bType == BT_RAW) {
            if (srcSize > dstCapacity) return mkerr(E_dstSize_tooSmall);       // ZSTD_copyRawBlock (zstd_decompressf.pas:699-713)
            rSize = srcSize;
            if (!window_room(b, rSize)) return mkerr(E_memory_allocation);
            if (rSize && hipMemcpy(b->state.dBytes + b->state.size, src, rSize, hipMemcpyHostToDevice) != hipSuccess) return mkerr(E_generic);
            if (rSize) memcpy(dst, src, rSize);
            b->expected -= rSize;
        } else {
            if (b->rleSize > dstCapacity) return mkerr(E_dstSize_tooSmall);    // ZSTD_setRleBlock (:715-731)
            rSize = b->rleSize;
            b->expected = 0;
            if (rSize > b->blockSizeMax) return mkerr(E_corruption);
            if (!window_room(b, rSize)) return mkerr(E_memory_allocation);
            if (rSize && hipMemset(b->state.dBytes + b->state.size, src[0], rSize) != hipSuccess) return mkerr(E_generic);
            if (rSize) memset(dst, src[0], rSize);
        }
        if (rSize > b->blockSizeMax) return mkerr(E_corruption);
        b->decodedSize += rSize;
        if (b->validateChecksum && rSize) launch_xxh64_stream(b->dXxh, b->state.dBytes + b->state.size, rSize, nullptr);
        b->state.size += (uint32_t)rSize;
        if (b->expected > 0) return rSize;      // the rest of a raw block is still to come
        if (b->stage == BL::decompressLastBlock) {
            if (b->fh.contentSize != ZSTDGPU_CONTENTSIZE_UNKNOWN && b->decodedSize != b->fh.contentSize) return mkerr(E_corruption);
            if (b->fh.checksumFlag) { b->expected = 4; b->stage = BL::checkChecksum; }
            else { b->expected = 0; b->stage = BL::getFrameHeaderSize; }
        } else { b->stage = BL::decodeBlockHeader; b->expected = 3; }
        return rSize;
    }
    case BL::checkChecksum:
        if (b->validateChecksum) {
            uint64_t digest = 0;
            if (hipMemcpy(&digest, &b->dXxh->digest, 8, hipMemcpyDeviceToHost) != hipSuccess) return mkerr(E_generic);
            if (rd32(src) != (uint32_t)digest) return mkerr(E_checksum_wrong);
        }
        b->expected = 0;
        b->stage = BL::getFrameHeaderSize;
        return 0;
    case BL::decodeSkippableHeader:
        memcpy(b->headerBuffer + (8 - srcSize), src, srcSize);
        b->expected = rd32(b->headerBuffer + 4);
        b->stage = BL::skipFrame;
        return 0;
    case BL::skipFrame:
        b->expected = 0;
        b->stage = BL::getFrameHeaderSize;
        return 0;
    }
    return mkerr(E_generic);
}
catch (...) { return mkerr(E_memory_allocation); }

// ZSTD_decompress creates and frees its own context per call (zstd_decompressf.pas:977-988)
extern "C" size_t zstdgpu_decompress(void* dst, size_t dstCapacity, const void* src, size_t srcSize)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return mkerr(E_generic);
    zstdgpu_ctx* ctx = zstdgpu_create(dev);
    if (!ctx) return mkerr(E_memory_allocation);
    ctx->oneShotFixed = false;      // (a context that lives for one call: a scratch area of its own would cost more than it saves)
    size_t const r = decompress_multi(ctx, dst, dstCapacity, src, srcSize, nullptr);
    zstdgpu_destroy(ctx);
    return r;
}
