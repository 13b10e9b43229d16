// The context and dictionary objects behind the opaque handles of include/zstdgpu.h, shared by the source files
// of the host library.
#pragma once
#include <stdint.h>

#include <vector>

#include "../../include/zstdgpu.h"
#include "zstdgpu_internal.h"

struct zstdgpu_dstream;
struct zstdgpu_bufferless;

struct zstdgpu_ctx {
    int device = 0;
    // grow-only device staging for the host-buffer drop-ins
    uint8_t *dSrc = nullptr, *dDst = nullptr;
    size_t dSrcCap = 0, dDstCap = 0;
    // device buffers of destroyed batches, kept for the next batch (hipMalloc of the hundreds of MB of
    // literal/sequence scratch costs tens of milliseconds; a steady stream of batches should not pay it again)
    struct Spare { void* p; size_t bytes; };
    std::vector<Spare> spare;
    // decoder parameters (ZSTD_DCtx_resetParameters, zstd_decompressf.pas:131-139) and the referenced dictionary
    uint32_t format = 0;                          // ZSTD_f_zstd1 / ZSTD_f_zstd1_magicless
    uint64_t maxWindowSize = (1ull << 27) + 1;    // ZSTD_MAXWINDOWSIZE_DEFAULT (:51)
    uint32_t stableOut = 0, ignoreChecksum = 0;
    const zstdgpu_ddict* refDDict = nullptr;
    std::vector<zstdgpu_dstream*> streams;        // streams opened on this context
    std::vector<zstdgpu_ddict*> ddicts;           // dictionaries created on it (they forget it when it is destroyed)
    std::vector<zstdgpu_batch*> batches;          // batches created on it (likewise)
    uint8_t* scratch = nullptr;                   // this device's fixed scratch array (zstdgpu_decode_frames_noalloc)
    // the one-shot drop-ins decode small inputs inside a fixed scratch area of the context's own (allocated at the first
    // such call), with the frame list and the verdicts in a small buffer beside it: nothing is allocated per call
    uint8_t* oneShotScratch = nullptr;
    uint8_t* oneShotIO = nullptr;                 // 256 frame entries, then 256 verdicts
    bool oneShotTried = false;
    bool oneShotFixed = true;                     // false for the context that zstdgpu_decompress makes for a single call
    struct zstdgpu_bufferless* bufferless = nullptr;   // state of zstdgpu_decompressBegin / zstdgpu_decompressContinue (made at the first Begin)
    uint32_t execKernel = 0;                      // zstdgpu_test_setExecuteKernel: 0 chosen per launch, 1 one wave per frame, 2 eight
    uint32_t opts() const { return (format ? zg::kOptMagicless : 0u) | (ignoreChecksum ? zg::kOptIgnoreChecksum : 0u); }
    void reset_parameters() { format = 0; maxWindowSize = (1ull << 27) + 1; stableOut = 0; ignoreChecksum = 0; }
};

struct zstdgpu_ddict {
    zstdgpu_ctx* ctx = nullptr;
    uint8_t* dBytes = nullptr;     // whole dictionary on the device
    zg::DevDict* dDict = nullptr;
    uint32_t size = 0, dictID = 0, hasEntropy = 0, entropySize = 0, status = 0;
    uint32_t entropyMask = 0;      // which tables the dictionary brings: bit 0 Huffman, bit 1 the three sequence tables (3 for a digested dictionary)
    int device = 0;
};


namespace zg {
// This device's copy of the fixed scratch array of the no-allocation entry point (noalloc.hip); nullptr on failure.
uint8_t* noalloc_scratch(int device);
// The body of zstdgpu_decode_frames_noalloc on a scratch area of fixed_scratch_bytes() given by the caller (noalloc.hip).
size_t fixed_scratch_bytes();
// carry: when set, the entropy state that the first block of frame 0 leaves behind is moved into it after the chain
// (k_carry_entropy; zstdgpu_decompressContinue decodes one block per call this way).
size_t decode_frames_fixed(zstdgpu_ctx* ctx, uint8_t* scratch, const void* devSrc, size_t srcBytes, const zstdgpu_frame_in* devFrames, uint32_t nFrames,
                           void* devDst, zstdgpu_frame_status* devStatus, const zstdgpu_ddict* ddict, void* stream, DevDict* carry = nullptr);
}  // namespace zg
