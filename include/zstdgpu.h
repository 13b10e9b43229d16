/*
 * zstdgpu.h -- C ABI of the MI355X batched zstd decompressor.
 *
 * Drop-in boundary for the one-shot decode path of the Free Pascal zstd port
 * (terrylao/zstd): every entry point below names the Pascal routine it replaces
 * (file:line relative to the reference tree).  Plain C types only; all functions are
 * `cdecl`, never throw, and never call back into the host program.
 *
 * Error convention: upstream zstd's -- a size_t result that is either a byte count or
 * (size_t)-code with code from ZSTD_ErrorCode (common/error_private.pas:6-36).  The
 * port's own ERROR()/ERR_isError (common/error_private.pas:44-51) are inverted and are
 * not reproduced; use zstdgpu_isError().
 */
#ifndef ZSTDGPU_H
#define ZSTDGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZSTDGPU_VERSION_NUMBER 10408u                /* decodes the zstd 1.4.8 format (common/zstd.pas:13-15) */
#define ZSTDGPU_CONTENTSIZE_UNKNOWN (0ULL - 1)       /* upstream sentinels; the port's 32-bit ones (common/zstd.pas:19-20) are a defect */
#define ZSTDGPU_CONTENTSIZE_ERROR   (0ULL - 2)
#define ZSTDGPU_BLOCKSIZE_MAX (1u << 17)             /* common/zstd.pas:12 */

typedef struct zstdgpu_ctx zstdgpu_ctx;       /* replaces ZSTD_DCtx (decompress/zstd_ddict.pas:109-170): one per GPU, single-owner */
typedef struct zstdgpu_ddict zstdgpu_ddict;   /* replaces ZSTD_DDict (decompress/zstd_ddict.pas:84-92): digested tables + content in HBM */
typedef struct zstdgpu_batch zstdgpu_batch;   /* a prepared batch: device descriptors + scratch, reusable across launches */

/* ---- error helpers: ZSTD_isError / ZSTD_getErrorCode / ZSTD_getErrorName (common/error_private.pas:38-41, :66-106) ---- */
unsigned    zstdgpu_isError(size_t code);
unsigned    zstdgpu_getErrorCode(size_t code);
const char* zstdgpu_getErrorName(size_t code);
unsigned    zstdgpu_versionNumber(void);

/* ---- context: ZSTD_createDCtx / ZSTD_freeDCtx (decompress/zstd_decompressf.pas:169, :183) ----
 * `device` is a HIP device ordinal.  Returns NULL when no usable GPU/extension is present
 * (there is no CPU fallback). */
zstdgpu_ctx* zstdgpu_create(int device);
void         zstdgpu_destroy(zstdgpu_ctx* ctx);

/* ---- host-side header queries (stay on the host, as in the reference) ----
 * ZSTD_getFrameContentSize (decompress/zstd_decompressf.pas:414), ZSTD_findFrameCompressedSize (:646),
 * ZSTD_getDictID_fromDict (:1455), ZSTD_getDictID_fromFrame (:1473). */
unsigned long long zstdgpu_getFrameContentSize(const void* src, size_t srcSize);
size_t             zstdgpu_findFrameCompressedSize(const void* src, size_t srcSize);
unsigned           zstdgpu_isFrame(const void* src, size_t srcSize);                      /* ZSTD_isFrame :223 */
size_t             zstdgpu_frameHeaderSize(const void* src, size_t srcSize);              /* ZSTD_frameHeaderSize :269 */
typedef struct {                                  /* ZSTD_frameHeader, common/zstd.pas:489-497 */
    unsigned long long frameContentSize;          /* ZSTDGPU_CONTENTSIZE_UNKNOWN when the header has none; skippable: its size field */
    unsigned long long windowSize;
    unsigned blockSizeMax;
    unsigned frameType;                           /* 0 ZSTD_frame, 1 ZSTD_skippableFrame */
    unsigned headerSize;
    unsigned dictID;
    unsigned checksumFlag;
} zstdgpu_frameHeader;
size_t             zstdgpu_getFrameHeader(zstdgpu_frameHeader* zfh, const void* src, size_t srcSize);   /* ZSTD_getFrameHeader :403; 0, bytes wanted, or error */
unsigned long long zstdgpu_getDecompressedSize(const void* src, size_t srcSize);          /* ZSTD_getDecompressedSize :525 */
unsigned long long zstdgpu_findDecompressedSize(const void* src, size_t srcSize);         /* ZSTD_findDecompressedSize :466 (all frames) */
unsigned long long zstdgpu_decompressBound(const void* src, size_t srcSize);              /* ZSTD_decompressBound :660 (all frames) */
unsigned           zstdgpu_getDictID_fromDict(const void* dict, size_t dictSize);
unsigned           zstdgpu_getDictID_fromFrame(const void* src, size_t srcSize);

/* ---- dictionaries: ZSTD_createDDict / ZSTD_freeDDict (decompress/zstd_ddict.pas:294-326) ----
 * Copies the dictionary to the GPU and digests its entropy tables there once
 * (ZSTD_loadDEntropy, decompress/zstd_decompressf.pas:1261-1356).  NULL on allocation failure;
 * a dictionary with a corrupted entropy section yields a ddict whose use returns dictionary_corrupted. */
zstdgpu_ddict* zstdgpu_createDDict(zstdgpu_ctx* ctx, const void* dict, size_t dictSize);
void           zstdgpu_freeDDict(zstdgpu_ddict* ddict);

/* ---- one-shot drop-ins, host buffers in and out ----
 * ZSTD_decompress (decompress/zstd_decompressf.pas:977), ZSTD_decompressDCtx (:971),
 * ZSTD_decompress_usingDict (:944), ZSTD_decompress_usingDDict (:1496).
 * `src` may hold several concatenated and skippable frames (ZSTD_decompressMultiFrame, :836-942). */
size_t zstdgpu_decompress(void* dst, size_t dstCapacity, const void* src, size_t srcSize);
size_t zstdgpu_decompressDCtx(zstdgpu_ctx* ctx, void* dst, size_t dstCapacity, const void* src, size_t srcSize);
size_t zstdgpu_decompress_usingDict(zstdgpu_ctx* ctx, void* dst, size_t dstCapacity, const void* src, size_t srcSize,
                                    const void* dict, size_t dictSize);
size_t zstdgpu_decompress_usingDDict(zstdgpu_ctx* ctx, void* dst, size_t dstCapacity, const void* src, size_t srcSize,
                                     const zstdgpu_ddict* ddict);

/* ---- descriptor level: what the host-side block loop emits instead of decoding ----
 * The frame loop of ZSTD_decompressFrame (decompress/zstd_decompressf.pas:734-834) keeps parsing the
 * frame header (ZSTD_getFrameHeader_advanced, :281-395) and each 3-byte block header
 * (ZSTD_getcBlockSize, decompress/zstd_decompress_block.pas:190-207); instead of calling
 * ZSTD_decompressBlock_internal / ZSTD_copyRawBlock / ZSTD_setRleBlock it appends one
 * zstdgpu_block_desc per block.  Offsets are relative to one source buffer that exists both in
 * host memory (headers are read there) and in device memory (payloads are read there). */
typedef struct {
    uint64_t srcOff;        /* first byte of the frame (its magic number) */
    uint64_t srcSize;       /* bytes of this frame, checksum included */
    uint64_t dstOff;        /* where the regenerated bytes go, relative to the destination buffer */
    uint64_t dstCapacity;   /* room at dstOff */
    uint64_t contentSize;   /* frame content size or ZSTDGPU_CONTENTSIZE_UNKNOWN */
    uint64_t windowSize;
    uint32_t firstBlock;    /* index of the frame's first zstdgpu_block_desc */
    uint32_t nBlocks;
    uint32_t dictID;
    uint32_t checksumFlag;  /* 1: a 4-byte XXH64 trailer follows the last block */
    uint32_t checksum;      /* the trailer, read little-endian (valid when checksumFlag) */
    uint32_t hostError;     /* ZSTD_ErrorCode found while walking headers; the frame is then not decoded */
} zstdgpu_frame_desc;

typedef struct {
    uint64_t srcOff;        /* first byte after the 3-byte block header */
    uint32_t cSize;         /* payload bytes present in the source (1 for an RLE block) */
    uint32_t origSize;      /* RLE: regenerated size; raw: == cSize; compressed: unused */
    uint32_t frameIdx;
    uint8_t  blockType;     /* 0 raw, 1 RLE, 2 compressed (blockType_e, common/zstd_internal.pas:112-118) */
    uint8_t  lastBlock;
    uint16_t hostError;     /* ZSTD_ErrorCode hit at this point of the walk (e.g. reserved block type): decoding of the frame stops here */
} zstdgpu_block_desc;

typedef struct {
    uint32_t errorCode;     /* 0 or a ZSTD_ErrorCode */
    uint32_t xxh64Low32;    /* low 32 bits of XXH64(content, seed 0) when the frame carries a checksum */
    uint64_t regenSize;     /* bytes written at dstOff */
} zstdgpu_frame_status;

/* ---- streaming (ZSTD_decompressStream and friends, decompress/zstd_decompressf.pas:1510-1637, :1978-2359)
 * A host state machine in front of the one-shot path: input is collected until one whole frame is present (its
 * size is found by following the block headers as they arrive), the frame is decoded on the GPU, and its output is
 * handed over as room is given.  Return value as in the reference: 0 when a frame has been decoded and flushed
 * completely, a positive hint while more input or more room is needed, or an error (sticky until the stream is
 * re-initialised).  Skippable frames are consumed and regenerate nothing. */
typedef struct zstdgpu_dstream zstdgpu_dstream;
typedef struct { const void* src; size_t size; size_t pos; } zstdgpu_inBuffer;     /* ZSTD_inBuffer */
typedef struct { void* dst; size_t size; size_t pos; } zstdgpu_outBuffer;          /* ZSTD_outBuffer */
zstdgpu_dstream* zstdgpu_createDStream(zstdgpu_ctx* ctx);                          /* ZSTD_createDStream :1510 */
void   zstdgpu_freeDStream(zstdgpu_dstream* zds);                                  /* ZSTD_freeDStream :1526 */
size_t zstdgpu_initDStream(zstdgpu_dstream* zds);                                  /* ZSTD_initDStream :1605 */
size_t zstdgpu_initDStream_usingDDict(zstdgpu_dstream* zds, const zstdgpu_ddict* ddict);   /* :1614 */
size_t zstdgpu_resetDStream(zstdgpu_dstream* zds);                                 /* ZSTD_resetDStream :1630 */
size_t zstdgpu_decompressStream(zstdgpu_dstream* zds, zstdgpu_outBuffer* output, zstdgpu_inBuffer* input);   /* :1978 */

/* ---- bufferless streaming, one block per call (decompress/zstd_decompressf.pas:995-1246, :1386-1452)
 * ZSTD_decompressBegin :1386, ZSTD_decompressBegin_usingDict :1410, ZSTD_decompressBegin_usingDDict :1426,
 * ZSTD_nextSrcSizeToDecompress :995, ZSTD_nextInputType :1019, ZSTD_decompressContinue :1050.
 * The caller asks how many bytes come next, hands over exactly that many (a raw block may come in pieces), and gets the
 * bytes of every block back from the call that brought the block: 0 for header, checksum and skippable-frame calls, the
 * regenerated size for block calls, or an error.  The stage machine is the reference's.  A compressed block is decoded by
 * the GPU stage chain with the state the blocks before it left (entropy tables, repeat offsets, window) held on the device.
 * Host buffers in and out.  One frame at a time per context; after a frame's last call zstdgpu_nextSrcSizeToDecompress
 * returns 0 and Begin starts the next frame.  Differences from the reference: where the caller puts the output pieces does
 * not matter (the window is a device buffer of the context, not the caller's memory), and no more than the last 1.5 GB
 * of output stay reachable for matches whatever window the frame header states.  The context's format and checksum
 * parameters apply.  After an error the frame cannot be continued (Begin again). */
typedef enum {
    ZSTDGPU_nit_frameHeader = 0, ZSTDGPU_nit_blockHeader = 1, ZSTDGPU_nit_block = 2, ZSTDGPU_nit_lastBlock = 3,
    ZSTDGPU_nit_checksum = 4, ZSTDGPU_nit_skippableFrame = 5
} zstdgpu_nextInputType_e;                                                          /* ZSTD_nextInputType_e common/zstd.pas:584 */
size_t   zstdgpu_decompressBegin(zstdgpu_ctx* ctx);
size_t   zstdgpu_decompressBegin_usingDict(zstdgpu_ctx* ctx, const void* dict, size_t dictSize);
size_t   zstdgpu_decompressBegin_usingDDict(zstdgpu_ctx* ctx, const zstdgpu_ddict* ddict);
size_t   zstdgpu_nextSrcSizeToDecompress(zstdgpu_ctx* ctx);
unsigned zstdgpu_nextInputType(zstdgpu_ctx* ctx);
size_t   zstdgpu_decompressContinue(zstdgpu_ctx* ctx, void* dst, size_t dstCapacity, const void* src, size_t srcSize);

/* ---- decoder parameters (ZSTD_dParameter common/zstd.pas:289-309, :580-582; decompress/zstd_decompressf.pas:1641-1826)
 * As in the reference, where a stream is the decoding context itself, parameters and the referenced dictionary
 * belong to the context: they apply to zstdgpu_decompressDCtx, to batches built from frame entries
 * (zstdgpu_batch_create_from_frames / _on_device) and to every stream opened on the context.  They can only be
 * changed while no stream of the context is inside a frame (stage_wrong otherwise).
 *   windowLogMax        streams refuse frames whose window exceeds 1 << value (default 27; 0 selects the default)
 *   format              0 = ZSTD_f_zstd1, 1 = ZSTD_f_zstd1_magicless (frames start at the descriptor byte)
 *   stableOutBuffer     1: the caller passes the same, large enough, output buffer to every call of a frame
 *   forceIgnoreChecksum 1: the checksum trailer is skipped instead of compared */
typedef enum {
    ZSTDGPU_d_windowLogMax = 100, ZSTDGPU_d_format = 1000, ZSTDGPU_d_stableOutBuffer = 1001, ZSTDGPU_d_forceIgnoreChecksum = 1002
} zstdgpu_dParameter;
typedef enum {
    ZSTDGPU_reset_session_only = 1, ZSTDGPU_reset_parameters = 2, ZSTDGPU_reset_session_and_parameters = 3
} zstdgpu_ResetDirective;                                                           /* ZSTD_ResetDirective zstd.pas:275-279 */
typedef struct { size_t error; int lowerBound; int upperBound; } zstdgpu_bounds;   /* ZSTD_bounds zstd.pas:270-274 */
zstdgpu_bounds zstdgpu_dParam_getBounds(zstdgpu_dParameter param);                               /* :1680 */
size_t zstdgpu_DCtx_setParameter(zstdgpu_ctx* ctx, zstdgpu_dParameter param, int value);        /* :1763 */
size_t zstdgpu_DCtx_getParameter(zstdgpu_ctx* ctx, zstdgpu_dParameter param, int* value);       /* :1735 */
size_t zstdgpu_DCtx_refDDict(zstdgpu_ctx* ctx, const zstdgpu_ddict* ddict);                     /* :1641; used by zstdgpu_decompressDCtx (ZSTD_getDDict :951) */
size_t zstdgpu_DCtx_reset(zstdgpu_ctx* ctx, zstdgpu_ResetDirective reset);                      /* :1809 */

/* One source frame of a batch: where it is and where its output goes. */
typedef struct {
    uint64_t srcOff, srcSize, dstOff, dstCapacity;
} zstdgpu_frame_in;

/* C mirror of the host descriptor builder: walks the frame and block headers of each listed frame
 * (one zstd frame each) in host memory and fills the descriptor arrays.  `blocks` may be NULL to
 * count only.  Returns the number of block descriptors needed/written, or an error if
 * blockCapacity is too small (dstSize_tooSmall). */
size_t zstdgpu_buildDescriptors(const void* hostSrc, size_t srcBytes,
                                const zstdgpu_frame_in* in, uint32_t nFrames,
                                zstdgpu_frame_desc* frames,
                                zstdgpu_block_desc* blocks, size_t blockCapacity);

/* Prepare a batch from descriptors: parses literal/sequence section headers on the host
 * (the header switch of ZSTD_decodeLiteralsBlock, decompress/zstd_decompress_block.pas:243-274, :333-400,
 * and the count/modes bytes of ZSTD_decodeSeqHeaders, :708-742), links repeat-mode tables to the
 * block that defines them, sizes and allocates device scratch, uploads everything.
 * `ddict` may be NULL.  Returns NULL on allocation failure. */
zstdgpu_batch* zstdgpu_batch_create(zstdgpu_ctx* ctx, const void* hostSrc, size_t srcBytes,
                                    const zstdgpu_frame_desc* frames, uint32_t nFrames,
                                    const zstdgpu_block_desc* blocks, uint32_t nBlocks,
                                    const zstdgpu_ddict* ddict);
/* Convenience: zstdgpu_buildDescriptors + zstdgpu_batch_create. */
zstdgpu_batch* zstdgpu_batch_create_from_frames(zstdgpu_ctx* ctx, const void* hostSrc, size_t srcBytes,
                                                const zstdgpu_frame_in* in, uint32_t nFrames,
                                                const zstdgpu_ddict* ddict);
/* The same batch built from a source that is already in DEVICE memory: the frame headers
 * (ZSTD_getFrameHeader_advanced, decompress/zstd_decompressf.pas:281-395), the block headers (ZSTD_getcBlockSize,
 * decompress/zstd_decompress_block.pas:190-207; the loop of ZSTD_findFrameSizeInfo, decompress/zstd_decompressf.pas:566-639)
 * and the literal / sequence section headers (decompress/zstd_decompress_block.pas:243-274, :333-400, :708-742) are
 * walked by device kernels, one lane per frame; no host copy of the compressed bytes is needed.  `in` is a host array.
 * The descriptors are identical to those of the host walk.  Synchronous (two small device-to-host copies size the
 * scratch buffers).  `devSrc` must hold the same bytes when the batch is decoded.  Returns NULL on failure. */
zstdgpu_batch* zstdgpu_batch_create_on_device(zstdgpu_ctx* ctx, const void* devSrc, size_t srcBytes,
                                              const zstdgpu_frame_in* in, uint32_t nFrames,
                                              const zstdgpu_ddict* ddict);
/* Test hook: a 64-bit digest of the batch's descriptor arrays (blocks, frames, job lists); the host walk and the
 * device walk over the same input give the same value. */
uint64_t zstdgpu_batch_descriptorDigest(zstdgpu_batch* batch);
/* Destroy a batch before its context.  The two large scratch buffers (literals, sequence records) go back to the
 * context and are handed to the next batch that fits them, so a steady stream of batches allocates once. */
void zstdgpu_batch_destroy(zstdgpu_batch* batch);

/* Launch the whole stage chain (literals, sequence decode, layout, sequence execution, XXH64,
 * verdict) for the batch on `stream` (a hipStream_t, NULL = default stream).  `devSrc`/`devDst`
 * are device pointers to the source and destination buffers the descriptors index.
 * Asynchronous; allocates nothing.  Returns 0 or an error code (GENERIC on a launch failure). */
size_t zstdgpu_batch_decode(zstdgpu_batch* batch, const void* devSrc, void* devDst, void* stream);

/* ---- one entry point that allocates nothing ----
 * No counterpart in the reference.  It is for callers that cannot allocate device memory: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  It does what zstdgpu_batch_create_on_device +
 * zstdgpu_batch_decode do (device-side header walk, then the whole stage chain) with every intermediate buffer in one
 * fixed device array of ZSTDGPU_NOALLOC_SCRATCH_MIB MiB that each GPU holds from zstdgpu_create until the process
 * ends; it calls no hipMalloc / hipFree, copies nothing to or from the host and does not wait.
 *   devSrc, devFrames   DEVICE memory: the compressed frames and one zstdgpu_frame_in per frame (one zstd frame each)
 *   devDst, devStatus   DEVICE memory: the regenerated bytes and one zstdgpu_frame_status per frame, valid once the
 *                       stream has reached the end of the call's work
 * Limits: nFrames <= ZSTDGPU_NOALLOC_MAX_FRAMES and srcBytes <= ZSTDGPU_NOALLOC_MAX_SRC_BYTES, else the call returns
 * parameter_outOfBound and launches nothing.  What only the frames' contents tell is checked on the device: more than
 * ZSTDGPU_NOALLOC_MAX_BLOCKS blocks, ZSTDGPU_NOALLOC_MAX_LITERAL_BYTES Huffman-coded literal bytes or
 * ZSTDGPU_NOALLOC_MAX_SEQUENCES sequence records (sequences + 65 per block) in the batch: every frame of the call
 * gets errorCode memory_allocation and nothing is decoded.
 * All calls on one GPU share that GPU's array: calls that overlap in time on one GPU race, whatever their streams
 * and contexts; calls in one stream are ordered.  The context's format / checksum parameters apply. */
#define ZSTDGPU_NOALLOC_MAX_FRAMES 256u
#define ZSTDGPU_NOALLOC_MAX_SRC_BYTES (4u << 20)
#define ZSTDGPU_NOALLOC_MAX_BLOCKS 1024u
#define ZSTDGPU_NOALLOC_MAX_LITERAL_BYTES (4u << 20)
#define ZSTDGPU_NOALLOC_MAX_SEQUENCES (1u << 18)
#define ZSTDGPU_NOALLOC_SCRATCH_MIB 22u
size_t zstdgpu_decode_frames_noalloc(zstdgpu_ctx* ctx, const void* devSrc, size_t srcBytes,
                                     const zstdgpu_frame_in* devFrames, uint32_t nFrames,
                                     void* devDst, zstdgpu_frame_status* devStatus,
                                     const zstdgpu_ddict* ddict, void* stream);

/* Wait for the last launch and copy the per-frame verdicts out (`status` has nFrames entries). */
size_t zstdgpu_batch_status(zstdgpu_batch* batch, zstdgpu_frame_status* status);

/* Batch facts for measurement: totals over frames without a host error. */
uint32_t zstdgpu_batch_nFrames(const zstdgpu_batch* batch);
uint32_t zstdgpu_batch_nBlocks(const zstdgpu_batch* batch);
uint64_t zstdgpu_batch_srcBytes(const zstdgpu_batch* batch);      /* sum of frame srcSize (C) */
uint64_t zstdgpu_batch_scratchBytes(const zstdgpu_batch* batch);  /* device scratch held */
/* Which kernel variants the batch's shape selects (a fact of the descriptors, for tests and reports): bit 0 every sequence
 * job takes its three tables from the dictionary, bit 1 every literal job is a single stream coded with the dictionary's
 * Huffman table -- the launches then keep one copy of those tables per workgroup instead of one per block. */
uint32_t zstdgpu_batch_kernelVariants(const zstdgpu_batch* batch);

/* Per-stage timing of the next launches: when enabled, zstdgpu_batch_decode brackets each kernel
 * with HIP events on the launch stream; zstdgpu_batch_stage_ms (after zstdgpu_batch_status or a
 * stream sync) returns the milliseconds of the last launch, ZSTDGPU_NUM_STAGES floats. */
#define ZSTDGPU_NUM_STAGES 6
#define ZSTDGPU_STAGE_LITERALS 0   /* Huffman table build + literal streams */
#define ZSTDGPU_STAGE_SEQUENCES 1  /* FSE table build + sequence decode */
#define ZSTDGPU_STAGE_LAYOUT 2     /* per-frame block offsets and repeat-offset hand-over */
#define ZSTDGPU_STAGE_EXECUTE 3    /* sequence execution + raw/RLE blocks */
#define ZSTDGPU_STAGE_CHECKSUM 4   /* XXH64 + frame verdict */
#define ZSTDGPU_STAGE_TOTAL 5      /* first launch to last completion */
void   zstdgpu_batch_set_profiling(zstdgpu_batch* batch, int enabled);
size_t zstdgpu_batch_stage_ms(zstdgpu_batch* batch, float* ms);
const char* zstdgpu_stage_kernel_name(int stage);

/* ---- stage-level hooks, for checking single kernels against the units they replace ---- */
/* XXH64 of n device buffers (xxhash.pas:409-488): digests[i] = XXH64(devBase+off[i], len[i], seed 0). */
size_t zstdgpu_xxh64Batch(zstdgpu_ctx* ctx, const void* devBase, const uint64_t* off, const uint64_t* len,
                          uint32_t n, uint64_t* digests);
/* The three predefined sequence decoding tables as the sequence kernel builds them (the set_basic arm of
 * ZSTD_buildSeqTable, decompress/zstd_decompress_block.pas:626-688; the reference holds them as the constants
 * LL_defaultDTable / OF_defaultDTable / ML_defaultDTable, :29-125): cells[t][64][4] = {nextState, nbAdditionalBits, nbBits,
 * baseValue}, t = 0 LL (64 cells), 1 OF (32 cells, the rest zero), 2 ML (64 cells). */
size_t zstdgpu_predefinedTables(zstdgpu_ctx* ctx, uint32_t* cells);
/* Intermediates of the last launch, copied to host arrays (any may be NULL):
 * literals of compressed blocks as decoded by stage 0 (ZSTD_decodeLiteralsBlock), sequences as
 * decoded by stage 1 with repeat offsets resolved (ZSTD_decodeSequence), per block. */
size_t zstdgpu_batch_blockInfo(zstdgpu_batch* batch, uint32_t blockIdx,
                               uint32_t* litSize, uint32_t* nbSeq, uint64_t* regenSize, uint32_t* errorCode);
size_t zstdgpu_batch_copyLiterals(zstdgpu_batch* batch, const void* devSrc, uint32_t blockIdx, void* out, size_t outCapacity);
size_t zstdgpu_batch_copySequences(zstdgpu_batch* batch, uint32_t blockIdx, uint32_t* triples, size_t tripleCapacity);
/* Which of the two execute kernels takes the frames of every later launch of this context (batches and the
 * no-allocation entry point): 0 the library chooses by the shape of the batch (the default), 1 one wave per frame,
 * 2 eight waves per frame.  Both kernels reproduce ZSTD_execSequence (decompress/zstd_decompress_block.pas:896-1039)
 * and give the same bytes and verdicts; the knob exists so that tests and measurements can drive each of them over
 * the same input.  Returns 0, or parameter_outOfBound. */
size_t zstdgpu_test_setExecuteKernel(zstdgpu_ctx* ctx, int kernel);

#ifdef __cplusplus
}
#endif
#endif /* ZSTDGPU_H */
