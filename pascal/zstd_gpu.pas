{ Binding unit for libzstdgpu.so (include/zstdgpu.h): the C ABI of the batched gfx950 zstd decoder, as the Free Pascal
  port binds it.  Not compiled in the build image (it has no Free Pascal compiler); kept in step with the header by hand. }
unit zstd_gpu;
{$mode objfpc}{$H+}{$PACKRECORDS C}
interface
const
  ZSTDGPU_LIB = 'zstdgpu';
type
  pzstdgpu_ctx   = pointer;
  pzstdgpu_ddict = pointer;
  pzstdgpu_batch = pointer;
  zstdgpu_frame_desc = record          { mirrors include/zstdgpu.h }
    srcOff, srcSize, dstOff, dstCapacity, contentSize, windowSize: QWord;
    firstBlock, nBlocks, dictID, checksumFlag, checksum, hostError: LongWord;
  end;
  zstdgpu_block_desc = record
    srcOff: QWord;
    cSize, origSize, frameIdx: LongWord;
    blockType, lastBlock: Byte;
    hostError: Word;
  end;
  zstdgpu_frame_status = record
    errorCode, xxh64Low32: LongWord;
    regenSize: QWord;
  end;
  zstdgpu_frame_in = record            { one frame of a batch whose headers the library walks }
    srcOff, srcSize, dstOff, dstCapacity: QWord;
  end;
  zstdgpu_bounds = record              { ZSTD_bounds, common/zstd.pas:270-274 }
    error: PtrUInt;
    lowerBound, upperBound: LongInt;
  end;
  pzstdgpu_frame_in = ^zstdgpu_frame_in;
  pzstdgpu_frame_desc = ^zstdgpu_frame_desc;
  pzstdgpu_block_desc = ^zstdgpu_block_desc;
  pzstdgpu_frame_status = ^zstdgpu_frame_status;

{ results are PtrUInt with upstream's (size_t)-code convention; the port's ERR_isError
  (common/error_private.pas:48-51) must not be applied to them }
function zstdgpu_isError(code: PtrUInt): LongWord; cdecl; external ZSTDGPU_LIB;
function zstdgpu_getErrorCode(code: PtrUInt): LongWord; cdecl; external ZSTDGPU_LIB;
function zstdgpu_getErrorName(code: PtrUInt): PChar; cdecl; external ZSTDGPU_LIB;
function zstdgpu_versionNumber: LongWord; cdecl; external ZSTDGPU_LIB;   { 10408: decodes the zstd 1.4.8 format }

{ header / size queries: ZSTD_isFrame, ZSTD_frameHeaderSize, ZSTD_getFrameHeader, ZSTD_getDecompressedSize,
  ZSTD_findDecompressedSize, ZSTD_decompressBound (decompress/zstd_decompressf.pas:223, :269, :403, :525, :466, :660);
  zstdgpu_frameHeader has the fields of ZSTD_frameHeader (common/zstd.pas:489-497), frameType as LongWord }
function zstdgpu_getFrameContentSize(src: pbyte; srcSize: PtrUInt): QWord; cdecl; external ZSTDGPU_LIB;      { :414; QWord(-1) unknown, QWord(-2) error }
function zstdgpu_findFrameCompressedSize(src: pbyte; srcSize: PtrUInt): PtrUInt; cdecl; external ZSTDGPU_LIB; { :646 }
function zstdgpu_getDictID_fromDict(dict: pbyte; dictSize: PtrUInt): LongWord; cdecl; external ZSTDGPU_LIB;    { :1455 }
function zstdgpu_getDictID_fromFrame(src: pbyte; srcSize: PtrUInt): LongWord; cdecl; external ZSTDGPU_LIB;     { :1473 }
function zstdgpu_isFrame(src: pbyte; srcSize: PtrUInt): LongWord; cdecl; external ZSTDGPU_LIB;
function zstdgpu_frameHeaderSize(src: pbyte; srcSize: PtrUInt): PtrUInt; cdecl; external ZSTDGPU_LIB;
function zstdgpu_getFrameHeader(zfh: pointer; src: pbyte; srcSize: PtrUInt): PtrUInt; cdecl; external ZSTDGPU_LIB;
function zstdgpu_getDecompressedSize(src: pbyte; srcSize: PtrUInt): QWord; cdecl; external ZSTDGPU_LIB;
function zstdgpu_findDecompressedSize(src: pbyte; srcSize: PtrUInt): QWord; cdecl; external ZSTDGPU_LIB;
function zstdgpu_decompressBound(src: pbyte; srcSize: PtrUInt): QWord; cdecl; external ZSTDGPU_LIB;
function zstdgpu_create(device: LongInt): pzstdgpu_ctx; cdecl; external ZSTDGPU_LIB;
procedure zstdgpu_destroy(ctx: pzstdgpu_ctx); cdecl; external ZSTDGPU_LIB;
function zstdgpu_createDDict(ctx: pzstdgpu_ctx; dict: pbyte; dictSize: PtrUInt): pzstdgpu_ddict; cdecl; external ZSTDGPU_LIB;
procedure zstdgpu_freeDDict(ddict: pzstdgpu_ddict); cdecl; external ZSTDGPU_LIB;

{ drop-ins for decompress/zstd_decompressf.pas:977 (ZSTD_decompress), :971 (ZSTD_decompressDCtx),
  :944 (ZSTD_decompress_usingDict), :1496 (ZSTD_decompress_usingDDict) }
function zstdgpu_decompress(dst: pbyte; dstCapacity: PtrUInt; src: pbyte; srcSize: PtrUInt): PtrUInt; cdecl; external ZSTDGPU_LIB;
function zstdgpu_decompressDCtx(ctx: pzstdgpu_ctx; dst: pbyte; dstCapacity: PtrUInt; src: pbyte; srcSize: PtrUInt): PtrUInt; cdecl; external ZSTDGPU_LIB;
function zstdgpu_decompress_usingDict(ctx: pzstdgpu_ctx; dst: pbyte; dstCapacity: PtrUInt; src: pbyte; srcSize: PtrUInt; dict: pbyte; dictSize: PtrUInt): PtrUInt; cdecl; external ZSTDGPU_LIB;
function zstdgpu_decompress_usingDDict(ctx: pzstdgpu_ctx; dst: pbyte; dstCapacity: PtrUInt; src: pbyte; srcSize: PtrUInt; ddict: pzstdgpu_ddict): PtrUInt; cdecl; external ZSTDGPU_LIB;

{ descriptor level }
function zstdgpu_batch_create(ctx: pzstdgpu_ctx; hostSrc: pbyte; srcBytes: PtrUInt;
  frames: pzstdgpu_frame_desc; nFrames: LongWord; blocks: pzstdgpu_block_desc; nBlocks: LongWord;
  ddict: pzstdgpu_ddict): pzstdgpu_batch; cdecl; external ZSTDGPU_LIB;
{ the same batch from a source already in device memory: headers walked by device kernels; `frames_in` = array of
  record srcOff, srcSize, dstOff, dstCapacity: QWord end }
function zstdgpu_batch_create_on_device(ctx: pzstdgpu_ctx; devSrc: pointer; srcBytes: PtrUInt;
  frames_in: pointer; nFrames: LongWord; ddict: pzstdgpu_ddict): pzstdgpu_batch; cdecl; external ZSTDGPU_LIB;
{ the same from frame entries, headers walked by the library on the host copy of the source }
function zstdgpu_batch_create_from_frames(ctx: pzstdgpu_ctx; hostSrc: pbyte; srcBytes: PtrUInt;
  frames_in: pointer; nFrames: LongWord; ddict: pzstdgpu_ddict): pzstdgpu_batch; cdecl; external ZSTDGPU_LIB;
function zstdgpu_batch_decode(batch: pzstdgpu_batch; devSrc, devDst, stream: pointer): PtrUInt; cdecl; external ZSTDGPU_LIB;
function zstdgpu_batch_status(batch: pzstdgpu_batch; status: pzstdgpu_frame_status): PtrUInt; cdecl; external ZSTDGPU_LIB;
procedure zstdgpu_batch_destroy(batch: pzstdgpu_batch); cdecl; external ZSTDGPU_LIB;
{ batch facts and diagnostics: counts, bytes of source (C) and of device scratch, a digest of the descriptors (equal for
  the host and the device walk of the same input), per-stage milliseconds of the last decode when profiling is on
  (stage i runs the kernel zstdgpu_stage_kernel_name(i)) }
function zstdgpu_batch_nFrames(batch: pzstdgpu_batch): LongWord; cdecl; external ZSTDGPU_LIB;
function zstdgpu_batch_nBlocks(batch: pzstdgpu_batch): LongWord; cdecl; external ZSTDGPU_LIB;
function zstdgpu_batch_srcBytes(batch: pzstdgpu_batch): QWord; cdecl; external ZSTDGPU_LIB;
function zstdgpu_batch_scratchBytes(batch: pzstdgpu_batch): QWord; cdecl; external ZSTDGPU_LIB;
function zstdgpu_batch_descriptorDigest(batch: pzstdgpu_batch): QWord; cdecl; external ZSTDGPU_LIB;
function zstdgpu_batch_kernelVariants(batch: pzstdgpu_batch): LongWord; cdecl; external ZSTDGPU_LIB;   { bit 0 / bit 1: shared sequence tables / Huffman table }
procedure zstdgpu_batch_set_profiling(batch: pzstdgpu_batch; enabled: LongInt); cdecl; external ZSTDGPU_LIB;
function zstdgpu_batch_stage_ms(batch: pzstdgpu_batch; ms: PSingle): PtrUInt; cdecl; external ZSTDGPU_LIB;
function zstdgpu_stage_kernel_name(stage: LongInt): PChar; cdecl; external ZSTDGPU_LIB;
{ stage-level hooks that mirror the reference's units, for tests: XXH64 of device ranges (common/xxhash.pas:533-827), a
  block's section facts, its decoded literals (ZSTD_decodeLiteralsBlock) and its sequences as (litLength, matchLength,
  offset) triples (ZSTD_decodeSequence) }
function zstdgpu_xxh64Batch(ctx: pzstdgpu_ctx; devBase: pointer; off, len: PQWord; n: LongWord; hashes: PQWord): PtrUInt; cdecl; external ZSTDGPU_LIB;
function zstdgpu_batch_blockInfo(batch: pzstdgpu_batch; blockIdx: LongWord; litSize, nbSeq: PLongWord; regen: PQWord; err: PLongWord): PtrUInt; cdecl; external ZSTDGPU_LIB;
function zstdgpu_batch_copyLiterals(batch: pzstdgpu_batch; devSrc: pointer; blockIdx: LongWord; dst: pbyte; dstCapacity: PtrUInt): PtrUInt; cdecl; external ZSTDGPU_LIB;
function zstdgpu_batch_copySequences(batch: pzstdgpu_batch; blockIdx: LongWord; triples: PLongWord; tripleCapacity: PtrUInt): PtrUInt; cdecl; external ZSTDGPU_LIB;
{ streaming: ZSTD_createDStream / ZSTD_initDStream / ZSTD_decompressStream (decompress/zstd_decompressf.pas:1510, :1605, :1978);
  zstdgpu_inBuffer / zstdgpu_outBuffer have the layout of ZSTD_inBuffer / ZSTD_outBuffer with PtrUInt sizes }
function zstdgpu_createDStream(ctx: pzstdgpu_ctx): pointer; cdecl; external ZSTDGPU_LIB;
procedure zstdgpu_freeDStream(zds: pointer); cdecl; external ZSTDGPU_LIB;
function zstdgpu_initDStream(zds: pointer): PtrUInt; cdecl; external ZSTDGPU_LIB;
function zstdgpu_initDStream_usingDDict(zds: pointer; ddict: pzstdgpu_ddict): PtrUInt; cdecl; external ZSTDGPU_LIB;
function zstdgpu_resetDStream(zds: pointer): PtrUInt; cdecl; external ZSTDGPU_LIB;
function zstdgpu_decompressStream(zds: pointer; output, input: pointer): PtrUInt; cdecl; external ZSTDGPU_LIB;
{ bufferless streaming, one block per call: ZSTD_decompressBegin / _usingDict / _usingDDict, ZSTD_nextSrcSizeToDecompress,
  ZSTD_nextInputType, ZSTD_decompressContinue (decompress/zstd_decompressf.pas:1386, :1410, :1426, :995, :1019, :1050);
  zstdgpu_nextInputType returns ord(ZSTD_nextInputType_e) (common/zstd.pas:584) }
function zstdgpu_decompressBegin(ctx: pzstdgpu_ctx): PtrUInt; cdecl; external ZSTDGPU_LIB;
function zstdgpu_decompressBegin_usingDict(ctx: pzstdgpu_ctx; dict: pbyte; dictSize: PtrUInt): PtrUInt; cdecl; external ZSTDGPU_LIB;
function zstdgpu_decompressBegin_usingDDict(ctx: pzstdgpu_ctx; ddict: pzstdgpu_ddict): PtrUInt; cdecl; external ZSTDGPU_LIB;
function zstdgpu_nextSrcSizeToDecompress(ctx: pzstdgpu_ctx): PtrUInt; cdecl; external ZSTDGPU_LIB;
function zstdgpu_nextInputType(ctx: pzstdgpu_ctx): LongWord; cdecl; external ZSTDGPU_LIB;
function zstdgpu_decompressContinue(ctx: pzstdgpu_ctx; dst: pbyte; dstCapacity: PtrUInt; src: pbyte; srcSize: PtrUInt): PtrUInt; cdecl; external ZSTDGPU_LIB;
{ decoder parameters: ZSTD_DCtx_setParameter / getParameter / refDDict / reset, ZSTD_dParam_getBounds
  (decompress/zstd_decompressf.pas:1641-1826); param takes ord(ZSTD_dParameter) (100, 1000, 1001, 1002), reset takes
  ord(ZSTD_ResetDirective) }
function zstdgpu_dParam_getBounds(param: LongInt): zstdgpu_bounds; cdecl; external ZSTDGPU_LIB;
function zstdgpu_DCtx_setParameter(ctx: pzstdgpu_ctx; param, value: LongInt): PtrUInt; cdecl; external ZSTDGPU_LIB;
function zstdgpu_DCtx_getParameter(ctx: pzstdgpu_ctx; param: LongInt; value: pLongInt): PtrUInt; cdecl; external ZSTDGPU_LIB;
function zstdgpu_DCtx_refDDict(ctx: pzstdgpu_ctx; ddict: pzstdgpu_ddict): PtrUInt; cdecl; external ZSTDGPU_LIB;
function zstdgpu_DCtx_reset(ctx: pzstdgpu_ctx; reset: LongInt): PtrUInt; cdecl; external ZSTDGPU_LIB;
{ no counterpart in the reference: device walk + decode for callers that cannot allocate (section 5); everything the
  pointers name is device memory, devFrames an array of zstdgpu_frame_in, devStatus one zstdgpu_frame_status per frame }
function zstdgpu_decode_frames_noalloc(ctx: pzstdgpu_ctx; devSrc: pointer; srcBytes: PtrUInt; devFrames: pointer;
  nFrames: LongWord; devDst, devStatus: pointer; ddict: pzstdgpu_ddict; stream: pointer): PtrUInt; cdecl; external ZSTDGPU_LIB;
{ test hook: the predefined LL / OF / ML decoding tables as the device builds them, cells[3][64][4] of
  (nextState, nbAdditionalBits, nbBits, baseValue) -- the constants LL_defaultDTable, OF_defaultDTable, ML_defaultDTable
  (decompress/zstd_decompress_block.pas:29-125) }
function zstdgpu_predefinedTables(ctx: pzstdgpu_ctx; cells: PLongWord): PtrUInt; cdecl; external ZSTDGPU_LIB;
{ test / measurement knob: which of the two execute kernels takes the frames of later launches of this context
  (0 chosen by the library, 1 one wave per frame, 2 eight waves per frame); same bytes and verdicts either way }
function zstdgpu_test_setExecuteKernel(ctx: pzstdgpu_ctx; kernel: LongInt): PtrUInt; cdecl; external ZSTDGPU_LIB;
{ the C mirror of the descriptor builder in zstd_gpu_descriptors.pas (frames_in -> frame and block descriptors) }
function zstdgpu_buildDescriptors(hostSrc: pbyte; srcBytes: PtrUInt; frames_in: pzstdgpu_frame_in; nFrames: LongWord;
  frames: pzstdgpu_frame_desc; blocks: pzstdgpu_block_desc; blockCapacity: PtrUInt): PtrUInt; cdecl; external ZSTDGPU_LIB;
implementation
end.
