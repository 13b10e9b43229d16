// The one entry point of the library that allocates nothing: zstdgpu_decode_frames_noalloc.
//
// It has no counterpart in the reference.  It is zstdgpu_batch_create_on_device + zstdgpu_batch_decode for callers
// that cannot allocate device memory (code under HIP graph capture, tight serving loops, harnesses that forbid
// hipMalloc): the descriptors are built by the same device kernels, the stage chain is the same, but every buffer the
// batch object would allocate lies in one fixed device array of this file, cut up by fixed capacities, and the block
// and job counts stay on the device (the kernels read them there), so nothing is copied back and nothing waits.
//
// The array exists once per GPU and is shared by every call on that GPU: calls that overlap on one GPU race.  It keeps
// the previous call's contents; what must start at zero is cleared by the call itself.
#include <hip/hip_runtime.h>

#include <mutex>

#include "../../include/zstdgpu.h"
#include "zstdgpu_internal.h"
#include "zstdgpu_objects.h"

namespace zg {
namespace {

constexpr size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

// capacities (the limits documented in include/zstdgpu.h)
constexpr uint32_t kMaxFrames = ZSTDGPU_NOALLOC_MAX_FRAMES;
constexpr uint32_t kMaxBlocks = ZSTDGPU_NOALLOC_MAX_BLOCKS;
constexpr uint64_t kCapLit = ZSTDGPU_NOALLOC_MAX_LITERAL_BYTES;
constexpr uint64_t kCapTmp = 3 * kCapLit;                           // split-decode scratch: about 2 x literals + 8 KB per block
constexpr uint64_t kCapSeq = ZSTDGPU_NOALLOC_MAX_SEQUENCES;        // sequence records (16 bytes each)

// layout of the scratch array
constexpr size_t oTotals = 0;
constexpr size_t oSlots = oTotals + up256(8 * sizeof(uint64_t));
constexpr size_t oFrames = oSlots + up256(kMaxFrames * sizeof(uint32_t));
constexpr size_t oBlocks = oFrames + up256(kMaxFrames * sizeof(DevFrame));
constexpr size_t oState = oBlocks + up256(kMaxBlocks * sizeof(DevBlock));
constexpr size_t oFlags = oState + up256(kMaxBlocks * sizeof(DevBlockState));
constexpr size_t oLitJobs = oFlags + up256(kMaxBlocks);
constexpr size_t oSeqJobs = oLitJobs + up256(kMaxBlocks * sizeof(uint32_t));
constexpr size_t oDryList = oSeqJobs + up256(kMaxBlocks * sizeof(uint32_t));
constexpr size_t oPackTab = oDryList + up256((kMaxBlocks + 1) * sizeof(uint32_t));
constexpr size_t oLitBuf = oPackTab + up256(((size_t)kMaxBlocks * 33 + 1) * sizeof(uint4));
constexpr size_t oLitTmp = oLitBuf + up256(kCapLit);
constexpr size_t oSeqBuf = oLitTmp + up256(kCapTmp);
constexpr size_t kScratchBytes = oSeqBuf + up256(kCapSeq * 16);

static_assert(kScratchBytes <= ((size_t)ZSTDGPU_NOALLOC_SCRATCH_MIB << 20), "the scratch stays within its stated size");

}  // namespace
}  // namespace zg

// The scratch itself: device memory that every GPU the library runs on holds until the process ends.
static __device__ __align__(256) unsigned char scratch[zg::kScratchBytes];

namespace zg {

// Address of this device's copy of the scratch array (each GPU has its own copy at its own address); looked up once
// per device with that device current.  nullptr when the lookup fails.
uint8_t* noalloc_scratch(int device)
{
    constexpr int kMaxDevices = 64;
    static std::mutex mu;
    static uint8_t* ptrs[kMaxDevices] = {};
    static bool tried[kMaxDevices] = {};
    if (device < 0 || device >= kMaxDevices) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!tried[device]) {
        tried[device] = true;
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) return nullptr;
        if (cur != device && hipSetDevice(device) != hipSuccess) return nullptr;
        uint8_t* ptr = nullptr;
        hipError_t const e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
        if (cur != device) hipSetDevice(cur);
        ptrs[device] = (e == hipSuccess) ? ptr : nullptr;
    }
    return ptrs[device];
}

size_t fixed_scratch_bytes() { return kScratchBytes; }

// Device walk + stage chain inside a scratch area of fixed_scratch_bytes() at `ptr` (256-byte aligned): the body of the
// entry point below, also used by the one-shot drop-ins with a scratch area that belongs to their context.
size_t decode_frames_fixed(zstdgpu_ctx* ctx, uint8_t* ptr, const void* devSrc, size_t srcBytes, const zstdgpu_frame_in* devFrames, uint32_t nFrames,
                           void* devDst, zstdgpu_frame_status* devStatus, const zstdgpu_ddict* ddict, void* stream, DevDict* carry)
{
    auto mkerr = [](uint32_t code) { return (size_t)0 - (size_t)code; };
    if (!ctx || hipSetDevice(ctx->device) != hipSuccess) return mkerr(E_generic);
    if (nFrames > kMaxFrames || srcBytes > ZSTDGPU_NOALLOC_MAX_SRC_BYTES) return mkerr(E_parameter_outOfBound);
    if (nFrames == 0) return 0;
    if (!devSrc || !devFrames || !devDst || !devStatus) return mkerr(E_generic);
    if (ddict && ddict->status) return mkerr(ddict->status);
    if (!ptr) return mkerr(E_memory_allocation);
    hipStream_t const s = (hipStream_t)stream;

    uint64_t* const totals = (uint64_t*)(ptr + oTotals);
    uint32_t* const slots = (uint32_t*)(ptr + oSlots);
    DevFrame* const frames = (DevFrame*)(ptr + oFrames);
    DevBlock* const blocks = (DevBlock*)(ptr + oBlocks);
    DevBlockState* const state = (DevBlockState*)(ptr + oState);
    uint8_t* const flags = ptr + oFlags;
    uint32_t* const litJobs = (uint32_t*)(ptr + oLitJobs);
    uint32_t* const seqJobs = (uint32_t*)(ptr + oSeqJobs);

    ScratchCaps const caps{kMaxBlocks, kCapLit, kCapTmp, kCapSeq};
    uint32_t const batchDictID = ddict ? ddict->dictID : 0;
    uint32_t const opts = (ddict ? (uint32_t)kOptDict : 0u) | ctx->opts();
    static_assert(sizeof(zstdgpu_frame_in) == sizeof(FrameIn), "frame entry layout");
    static_assert(sizeof(zstdgpu_frame_status) == sizeof(DevFrameStatus), "status layout");

    // ---- descriptors, on the device (the previous call's are overwritten; block slots and states start at zero)
    launch_scan_frames((const uint8_t*)devSrc, srcBytes, (const FrameIn*)devFrames, nFrames, batchDictID, opts, frames, slots, totals, s, caps);
    // Zeroed by a kernel, not by memset nodes: the call is meant to be captured into a graph, and replayed from one a memset
    // node of a few bytes (the eight totals, the dry list's count) was seen not to take effect on the second launch
    // (ROCm 7.2, gfx950).  Block descriptors, block states and flags lie side by side.
    static_assert(oBlocks < oState && oState < oFlags && oFlags < oLitJobs && (oLitJobs - oBlocks) % 16 == 0, "one zero fill covers blocks, states and flags");
    launch_zero(ptr + oBlocks, oLitJobs - oBlocks, s);
    launch_zero(ptr + oDryList, 16, s);
    launch_fill_blocks((const uint8_t*)devSrc, srcBytes, (const FrameIn*)devFrames, nFrames, batchDictID, opts, ddict ? ddict->entropyMask : 0u,
                       frames, blocks, flags, kMaxBlocks, litJobs, seqJobs, totals, s, caps);

    // ---- the stage chain, grids sized for the capacities; the kernels take the real counts from `totals`
    BatchParams p{};
    p.src = (const uint8_t*)devSrc; p.dst = (uint8_t*)devDst;
    p.blocks = blocks; p.state = state; p.frames = frames; p.status = (DevFrameStatus*)devStatus;
    p.litBuf = ptr + oLitBuf; p.litTmp = ptr + oLitTmp; p.packTab = (uint4*)(ptr + oPackTab); p.seqBuf = (uint32_t*)(ptr + oSeqBuf);
    p.litJobs = litJobs; p.seqJobs = seqJobs; p.dryList = (uint32_t*)(ptr + oDryList);
    p.nLitJobs = kMaxBlocks; p.nSeqJobs = kMaxBlocks; p.nFrames = nFrames; p.nBlocks = kMaxBlocks;
    p.devTotals = totals;
    p.execMode = ctx->execKernel;
    if (ddict) {
        p.dict = ddict->dDict;
        p.dictContent = ddict->dBytes + ddict->entropySize;
        p.dictContentSize = ddict->size - ddict->entropySize;
    }
    launch_literals(p, s);
    launch_pack_literals(p, s);
    launch_sequences(p, s);
    launch_sequences_dry(p, s);
    launch_layout(p, s);
    launch_execute(p, s);
    launch_checksum(p, s);
    if (carry) launch_carry_entropy(p, carry, s);      // block-at-a-time decoding: what the block leaves to the next one
    return hipGetLastError() == hipSuccess ? 0 : mkerr(E_generic);
}

}  // namespace zg

extern "C" size_t zstdgpu_decode_frames_noalloc(zstdgpu_ctx* ctx, const void* devSrc, size_t srcBytes,
                                                const zstdgpu_frame_in* devFrames, uint32_t nFrames,
                                                void* devDst, zstdgpu_frame_status* devStatus,
                                                const zstdgpu_ddict* ddict, void* stream)
{
    // ctx->scratch: this device's copy of `scratch` (looked up by zstdgpu_create)
    return zg::decode_frames_fixed(ctx, ctx ? ctx->scratch : nullptr, devSrc, srcBytes, devFrames, nFrames, devDst, devStatus, ddict, stream);
}
