// Structures shared by the host library and the gfx950 kernels.
#pragma once
#include <stdint.h>

namespace zg {

// ZSTD_ErrorCode values (common/error_private.pas:6-36)
enum : uint32_t {
    E_none = 0, E_generic = 1, E_prefix_unknown = 10, E_frameParameter_unsupported = 14,
    E_frameParameter_windowTooLarge = 16, E_corruption = 20, E_checksum_wrong = 22,
    E_dictionary_corrupted = 30, E_dictionary_wrong = 32, E_parameter_unsupported = 40, E_parameter_outOfBound = 42,
    E_stage_wrong = 60, E_memory_allocation = 64,
    E_dstSize_tooSmall = 70, E_srcSize_wrong = 72, E_dstBuffer_wrong = 104, E_maxCode = 120
};

// options of the frame walk (host and device builders): a dictionary is attached; ZSTD_f_zstd1_magicless
// (zstd.pas:416-417); ZSTD_d_ignoreChecksum (the trailer is skipped, not compared: zstd_decompressf.pas:815-828)
enum : uint32_t { kOptDict = 1, kOptMagicless = 2, kOptIgnoreChecksum = 4 };

enum : int32_t { SRC_DICT = -1, SRC_NONE = -2 };
enum : uint8_t { BT_RAW = 0, BT_RLE = 1, BT_COMPRESSED = 2, BT_STOP = 3 };   // BT_STOP: walk ended here with an error
enum : uint8_t { SET_BASIC = 0, SET_RLE = 1, SET_COMPRESSED = 2, SET_REPEAT = 3 };  // symbolEncodingType_e (zstd_internal.pas:112-118)

constexpr uint32_t kBlockSizeMax = 1u << 17;
// bytes per scratch region of the split literal decode (32 per block: every stream in 8 parts): a part regenerates
// about litSize/32 bytes; twice that plus slack is allowed before the block falls back to whole-stream decoding
constexpr uint32_t litTmpPart(uint32_t litSize) { return ((litSize + 3) / 4 / 4 + 256 + 31) & ~31u; }
constexpr uint64_t litTmpBytes(uint32_t litSize) { return 32ull * litTmpPart(litSize) + 64; }     // + a place for stores that must go nowhere
constexpr uint32_t kSymbolicRep = 0x80000000u;   // sequence offset / rep value that refers to the block's incoming rep stack

// Per-block descriptor consumed by the kernels (built on the host).
struct DevBlock {
    uint64_t srcOff;       // block payload in the source buffer
    uint64_t litBufOff;    // decoded literals go here (litType compressed/repeat)
    uint64_t seqBufOff;    // first sequence slot of this block
    uint64_t litTmpOff;    // 4-stream Huffman literals: 32 regions of litTmpPart(litSize) bytes for the split decode
    uint32_t cSize;
    uint32_t origSize;     // raw/RLE regenerated size
    uint32_t frameIdx;
    uint32_t litSize;      // regenerated literal bytes
    uint32_t litCSize;     // literal payload bytes after the literals header
    uint32_t nbSeq;
    uint32_t seqHdrOff;    // payload offset of the first table description (after the modes byte)
    int32_t  hufSrc;       // block whose literals header carries the Huffman table in use; SRC_DICT / SRC_NONE
    int32_t  llSrc, ofSrc, mlSrc;   // block that defines each sequence table in use; SRC_DICT / SRC_NONE
    uint16_t hostLitErr;   // error found on the host before/inside the literals header
    uint16_t hostSeqErr;   // error found on the host in the sequences header
    uint8_t  blockType;
    uint8_t  lastBlock;
    uint8_t  litType;
    uint8_t  litHdrSize;
    uint8_t  singleStream;
    uint8_t  llType, ofType, mlType;
};

// Per-block results written by the kernels.
struct DevBlockState {
    uint64_t regen;        // stage 1: litSize + sum(matchLength); stage 2 fills it for raw/RLE/no-sequence blocks
    uint64_t dstOff;       // stage 2: offset of the block inside its frame's output
    uint32_t litErr;       // stage 0
    uint32_t seqErr;       // stage 1: table description / bitstream init errors
    uint32_t seqValid;     // stage 1: sequences decoded before the bitstream ran dry
    uint32_t tailErr;      // stage 1: bitstream not consumed exactly
    uint32_t repOut[3];    // stage 1: rep stack after the block (entries may be symbolic)
    uint32_t repIn[3];     // stage 2: rep stack entering the block (concrete)
    uint32_t err;          // stage 3: verdict for the block
    uint32_t pad;
};

struct DevFrame {
    uint64_t dstOff, dstCapacity, contentSize;
    uint32_t firstBlock, nBlocks;
    uint32_t checksumFlag, checksum;
    uint32_t hostError;
    uint32_t useDict;      // 1: the batch dictionary applies to this frame
    uint32_t longWindow;   // 1: window above 2^24 -- zstd 1.4.8 may run its prefetching sequence loop on this frame's blocks
    uint32_t pad;          // (zero: the descriptor digest covers every byte)
};
static_assert(sizeof(DevFrame) == 56, "no implicit padding in DevFrame");

struct DevFrameStatus {    // same layout as zstdgpu_frame_status
    uint32_t errorCode, xxh64Low32;
    uint64_t regenSize;
};

// Digested dictionary in device memory (ZSTD_entropyDTables_t image in this library's cell formats).
constexpr uint32_t kHufCells = 1u << 12;      // HUF_TABLELOG_MAX
struct DevDict {
    uint32_t status;        // 0 ok, else ZSTD_ErrorCode (dictionary_corrupted)
    uint32_t hasEntropy;    // 1: tables below are valid (magic dictionary); 0: raw-content dictionary
    uint32_t entropySize;   // bytes before the content
    uint32_t hufLog;
    uint32_t llLog, ofLog, mlLog;
    uint32_t rep[3];
    uint32_t hufX1;         // 1: the Huffman table was built by the single-symbol decoder (a block's own table carried to the next block,
                            // k_carry_entropy); 0: by the double-symbol one, as a dictionary's always is (zstd_decompressf.pas:1288)
    uint32_t entropyValid;  // carried state only: bit 0 the Huffman table is set (litEntropy), bit 1 the sequence tables are (fseEntropy)
    uint16_t huf[kHufCells];   // {byte, nbBits<<8}
    uint32_t ll[512], of[256], ml[512];   // packed sequence cells
};
// (the kernels copy these tables 16 bytes at a time)
static_assert(offsetof(DevDict, huf) % 16 == 0 && offsetof(DevDict, ll) % 16 == 0 && offsetof(DevDict, of) % 16 == 0 && offsetof(DevDict, ml) % 16 == 0,
              "dictionary tables are 16-byte aligned");

// XXH64 of output that arrives in pieces (XXH64_state_t, xxhash.pas:57-66), with the digest of everything so far
struct DevXxhState {
    uint64_t v[4];
    uint64_t total;
    uint64_t digest;
    uint32_t memsize, pad;
    uint8_t mem[32];
};

struct BatchParams {
    const uint8_t* src;          // device source buffer
    uint8_t* dst;                // device destination buffer
    const DevBlock* blocks;
    DevBlockState* state;
    const DevFrame* frames;
    DevFrameStatus* status;
    uint8_t* litBuf;
    uint4* packTab;              // split literal decode: per literal job 16 parts {start, end, source address} + a flag
    uint8_t* litTmp;             // scratch of the split literal decode (may be null: streams are then decoded whole)
    uint32_t* seqBuf;            // triples (litLength, matchLength, offset)
    const uint32_t* litJobs;     // block indices with Huffman-coded literals
    const uint32_t* seqJobs;     // block indices with nbSeq > 0
    uint32_t* dryList;           // [0] count (cleared before every launch), then the blocks whose sequence stream ran dry early
    const DevDict* dict;         // may be null
    const uint8_t* dictContent;  // device pointer to dictionary content
    uint32_t dictContentSize;
    uint32_t nLitJobs, nSeqJobs, nFrames, nBlocks;
    uint32_t litDictSingleStream;  // 1: every literal job of the batch is a single stream coded with the dictionary's table (known to the host)
    uint32_t seqTablesFromDict;  // 1: every sequence job of the batch takes its three tables from the dictionary (known to the host)
    uint32_t execMode;           // which execute kernel takes a frame: 0 by its shape, 1 always one wave, 2 always eight (the caller's wish on
                                 // entry to launch_execute, which settles the 0 case by the size of the batch)
    // When set, the block and job counts are read from these device totals (layout of the descriptor builder's
    // `totals`) instead of the fields above, which then only bound the grids: the descriptors were built on the
    // device and nothing came back to the host.  totals[7] != 0: the batch does not fit its scratch, nothing is decoded.
    const uint64_t* devTotals;
#ifdef ZG_STAMPS
    uint64_t* dbg;               // diagnostic build only: 4 cycle counters per block
#endif
};

// One source frame of a batch (same layout as zstdgpu_frame_in).
struct FrameIn { uint64_t srcOff, srcSize, dstOff, dstCapacity; };

// Capacities of a fixed scratch area (all zero: no limit, the host sizes the buffers from the totals).  With limits the
// builder compares its totals with them on the device and raises totals[7] instead of reporting back to the host.
struct ScratchCaps { uint32_t maxBlocks; uint64_t litBytes, tmpBytes, seqRecords; };

// descriptor builder on the device (descriptors.hip).  totals: [0] block slots, [1] literal buffer bytes,
// [2] split-decode scratch bytes, [3] sequence records, [4] literal jobs, [5] sequence jobs, [6] source bytes of the
// frames without a header error, [7] the batch exceeds the scratch capacities, [8] sequence jobs with a table that is not the
// dictionary's, [9] literal jobs other than a single stream coded with the dictionary's table (16 entries, the rest unused)
void launch_scan_frames(const uint8_t* src, uint64_t srcBytes, const FrameIn* in, uint32_t nFrames, uint32_t batchDictID, uint32_t opts,
                        DevFrame* frames, uint32_t* slots, uint64_t* totals, void* stream, ScratchCaps caps = ScratchCaps{0, 0, 0, 0});
void launch_fill_blocks(const uint8_t* src, uint64_t srcBytes, const FrameIn* in, uint32_t nFrames, uint32_t batchDictID, uint32_t opts,
                        uint32_t dictEntropy, const DevFrame* frames, DevBlock* blocks, uint8_t* flags, uint32_t nBlocks,
                        uint32_t* litJobs, uint32_t* seqJobs, uint64_t* totals, void* stream, ScratchCaps caps = ScratchCaps{0, 0, 0, 0});

// kernel launchers (kernels.hip)
void launch_literals(const BatchParams& p, void* stream);
void launch_pack_literals(const BatchParams& p, void* stream);
void launch_sequences(const BatchParams& p, void* stream);
void launch_sequences_dry(const BatchParams& p, void* stream);
void launch_layout(const BatchParams& p, void* stream);
void launch_execute(const BatchParams& p, void* stream);
void launch_checksum(const BatchParams& p, void* stream);
void launch_digest_dict(const uint8_t* devDict, uint32_t dictSize, DevDict* out, void* stream);
void launch_basic_tables(uint32_t* out /* [3][64][4] */, void* stream);
void launch_zero(void* p /* 16-byte aligned */, size_t bytes /* multiple of 16 */, void* stream);
void launch_xxh64(const uint8_t* base, const uint64_t* off, const uint64_t* len, uint32_t n, uint64_t* out, void* stream);
void launch_carry_entropy(const BatchParams& p, DevDict* carry, void* stream);
void launch_xxh64_stream(DevXxhState* state, const uint8_t* data, uint64_t len, void* stream);

}  // namespace zg
