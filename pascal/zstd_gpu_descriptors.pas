{ Descriptor builder for the batched GPU decoder, in Pascal: what ZSTD_decompressFrame's block loop
  (decompress/zstd_decompressf.pas:734-808) becomes when the block bodies go to the GPU.  For every frame of a batch it
  parses the frame header (ZSTD_getFrameHeader_advanced, :281-395), follows the block headers (ZSTD_getcBlockSize,
  decompress/zstd_decompress_block.pas:190-207; the loop of ZSTD_findFrameSizeInfo, :566-639) and reads the checksum
  trailer (:815-828), and emits one zstdgpu_frame_desc per frame and one zstdgpu_block_desc per block -- the input of
  zstdgpu_batch_create.  It is the twin of zstdgpu_buildDescriptors in the library (zstd_amd/csrc/zstdgpu_host.cpp,
  walk_frame), which is what the tests drive; the two must produce the same arrays.

  Self-contained on purpose: it reads bytes itself instead of calling the port's MEM_readLE16/32/64, whose bodies are
  compiled out on little-endian targets by a misspelt conditional (common/zstd.pas:635).
  Not compiled in the build image (no Free Pascal compiler there). }
unit zstd_gpu_descriptors;
{$mode objfpc}{$H+}{$PACKRECORDS C}
interface
uses zstd_gpu;

const
  { ZSTD_ErrorCode values used here (common/error_private.pas:6-36) }
  ZG_E_prefix_unknown = 10;  ZG_E_frameParameter_unsupported = 14;  ZG_E_frameParameter_windowTooLarge = 16;
  ZG_E_corruption_detected = 20;  ZG_E_dstSize_tooSmall = 70;  ZG_E_srcSize_wrong = 72;
  ZG_BT_RAW = 0;  ZG_BT_RLE = 1;  ZG_BT_COMPRESSED = 2;  ZG_BT_STOP = 3;   { 3: the walk ended at this block with an error }
  ZG_CONTENTSIZE_UNKNOWN = QWord($FFFFFFFFFFFFFFFF);

{ Fills frames[0..nFrames-1] and, when blocks <> nil, blocks[0..result-1].  Returns the number of block descriptors
  (call once with blocks = nil to size the array), or QWord(-ZG_E_dstSize_tooSmall) when blockCapacity is too small. }
function ZSTDGPU_buildDescriptorsPas(src: pbyte; srcBytes: QWord; frames_in: pzstdgpu_frame_in; nFrames: LongWord;
  frames: pzstdgpu_frame_desc; blocks: pzstdgpu_block_desc; blockCapacity: QWord): QWord;

implementation

function rd16(p: pbyte): LongWord; inline; begin result := LongWord(p[0]) or (LongWord(p[1]) shl 8); end;
function rd24(p: pbyte): LongWord; inline; begin result := rd16(p) or (LongWord(p[2]) shl 16); end;
function rd32(p: pbyte): LongWord; inline; begin result := rd24(p) or (LongWord(p[3]) shl 24); end;
function rd64(p: pbyte): QWord; inline; begin result := QWord(rd32(p)) or (QWord(rd32(p + 4)) shl 32); end;

{ ZSTD_frameHeaderSize_internal (:246-279) for the zstd1 format; needs 5 bytes }
function frameHeaderSize(p: pbyte): QWord;
const fcs: array[0..3] of QWord = (0, 2, 4, 8);  did: array[0..3] of QWord = (0, 1, 2, 4);
var fhd: Byte;  single, fcsId: LongWord;
begin
  fhd := p[4];
  single := (fhd shr 5) and 1;
  fcsId := fhd shr 6;
  result := 5 + QWord(1 - single) + did[fhd and 3] + fcs[fcsId];
  if (single <> 0) and (fcsId = 0) then inc(result);
end;

{ one frame: header, block headers, trailer.  Appends block descriptors at index nBlocks (counting even when
  blocks = nil).  Returns the bytes of the frame that were consumed. }
function walkFrame(base: pbyte; const fin: zstdgpu_frame_in; frameIdx: LongWord; var fd: zstdgpu_frame_desc;
  blocks: pzstdgpu_block_desc; blockCapacity: QWord; var nBlocks: QWord): QWord;
var
  p: pbyte;  remaining, fhsize, ip, pos: QWord;
  fhd, wl: Byte;  single, fcsId, hdr, btype, bsize, cSize, windowLog: LongWord;
  bd: zstdgpu_block_desc;  done, stopped: Boolean;
  checksumFlag: LongWord;

  procedure fail(code: LongWord);
  begin
    fd.hostError := code;  fd.nBlocks := 0;  nBlocks := fd.firstBlock;
  end;

begin
  result := 0;
  p := base + fin.srcOff;
  remaining := fin.srcSize;
  FillChar(fd, SizeOf(fd), 0);
  fd.srcOff := fin.srcOff;  fd.srcSize := fin.srcSize;  fd.dstOff := fin.dstOff;  fd.dstCapacity := fin.dstCapacity;
  fd.contentSize := ZG_CONTENTSIZE_UNKNOWN;
  fd.firstBlock := LongWord(nBlocks);
  if remaining < 5 + 1 + 3 then begin fail(ZG_E_srcSize_wrong); exit; end;        { ZSTD_FRAMEHEADERSIZE_MIN + a block header (:755) }
  if rd32(p) <> $FD2FB528 then begin fail(ZG_E_prefix_unknown); exit; end;         { ZSTD_MAGICNUMBER, common/zstd.pas:5 }
  fhsize := frameHeaderSize(p);
  if remaining < fhsize + 3 then begin fail(ZG_E_srcSize_wrong); exit; end;
  fhd := p[4];
  if (fhd and $08) <> 0 then begin fail(ZG_E_frameParameter_unsupported); exit; end;   { reserved bit (:332) }
  single := (fhd shr 5) and 1;  fcsId := fhd shr 6;  checksumFlag := (fhd shr 2) and 1;
  pos := 5;
  if single = 0 then begin
    wl := p[pos];  inc(pos);
    windowLog := (wl shr 3) + 10;
    if windowLog > 31 then begin fail(ZG_E_frameParameter_windowTooLarge); exit; end;   { ZSTD_WINDOWLOG_MAX (:337-343) }
    fd.windowSize := QWord(1) shl windowLog;
    fd.windowSize := fd.windowSize + (fd.windowSize shr 3) * (wl and 7);
  end;
  case fhd and 3 of                                                                   { dictionary ID (:364-370) }
    1: begin fd.dictID := p[pos]; inc(pos, 1); end;
    2: begin fd.dictID := rd16(p + pos); inc(pos, 2); end;
    3: begin fd.dictID := rd32(p + pos); inc(pos, 4); end;
  end;
  case fcsId of                                                                       { frame content size (:371-383) }
    0: if single <> 0 then fd.contentSize := p[pos];
    1: fd.contentSize := QWord(rd16(p + pos)) + 256;
    2: fd.contentSize := rd32(p + pos);
    3: fd.contentSize := rd64(p + pos);
  end;
  if single <> 0 then fd.windowSize := fd.contentSize;
  { (a dictionary ID that is not the batch dictionary's is reported by zstdgpu_batch_create as dictionary_wrong) }

  ip := fhsize;
  dec(remaining, fhsize);
  stopped := false;
  repeat
    FillChar(bd, SizeOf(bd), 0);
    bd.frameIdx := frameIdx;
    done := false;
    if remaining < 3 then begin
      bd.blockType := ZG_BT_STOP;  bd.hostError := ZG_E_srcSize_wrong;  bd.lastBlock := 1;  done := true;  stopped := true;
    end else begin
      hdr := rd24(p + ip);                                                            { ZSTD_getcBlockSize (block:190-207) }
      btype := (hdr shr 1) and 3;  bsize := hdr shr 3;
      if btype = ZG_BT_RLE then cSize := 1 else cSize := bsize;
      if btype = 3 then begin
        bd.blockType := ZG_BT_STOP;  bd.hostError := ZG_E_corruption_detected;  bd.lastBlock := 1;  done := true;  stopped := true;
      end else if cSize > remaining - 3 then begin
        bd.blockType := ZG_BT_STOP;  bd.hostError := ZG_E_srcSize_wrong;  bd.lastBlock := 1;  done := true;  stopped := true;
      end else begin
        bd.srcOff := fin.srcOff + ip + 3;                                             { first byte after the block header }
        bd.cSize := cSize;  bd.origSize := bsize;  bd.blockType := Byte(btype);  bd.lastBlock := Byte(hdr and 1);
        inc(ip, 3 + QWord(cSize));  dec(remaining, 3 + QWord(cSize));
        done := bd.lastBlock <> 0;
      end;
    end;
    if blocks <> nil then begin
      if nBlocks < blockCapacity then blocks[nBlocks] := bd;
    end;
    inc(nBlocks);
  until done;
  fd.nBlocks := LongWord(nBlocks) - fd.firstBlock;
  if (not stopped) and (checksumFlag <> 0) then begin
    if remaining < 4 then fd.checksumFlag := 2                                        { trailer missing: checksum_wrong once all else passed (:817) }
    else begin fd.checksumFlag := 1;  fd.checksum := rd32(p + ip);  inc(ip, 4);  dec(remaining, 4); end;
  end;
  { a batch entry is exactly one frame: bytes behind it are an error of this entry }
  if (not stopped) and (fd.hostError = 0) and (fd.checksumFlag <> 2) and (ip <> fin.srcSize) then fd.hostError := ZG_E_srcSize_wrong;
  result := ip;
end;

function ZSTDGPU_buildDescriptorsPas(src: pbyte; srcBytes: QWord; frames_in: pzstdgpu_frame_in; nFrames: LongWord;
  frames: pzstdgpu_frame_desc; blocks: pzstdgpu_block_desc; blockCapacity: QWord): QWord;
var i: LongWord;  nBlocks: QWord;  fd: zstdgpu_frame_desc;
begin
  nBlocks := 0;
  if nFrames > 0 then
    for i := 0 to nFrames - 1 do begin
      if (frames_in[i].srcOff > srcBytes) or (frames_in[i].srcSize > srcBytes - frames_in[i].srcOff) then begin
        FillChar(fd, SizeOf(fd), 0);
        fd.srcOff := frames_in[i].srcOff;  fd.srcSize := frames_in[i].srcSize;
        fd.dstOff := frames_in[i].dstOff;  fd.dstCapacity := frames_in[i].dstCapacity;
        fd.firstBlock := LongWord(nBlocks);  fd.hostError := ZG_E_srcSize_wrong;  fd.contentSize := ZG_CONTENTSIZE_UNKNOWN;
      end else
        walkFrame(src, frames_in[i], i, fd, blocks, blockCapacity, nBlocks);
      if frames <> nil then frames[i] := fd;
    end;
  if (blocks <> nil) and (nBlocks > blockCapacity) then result := QWord(0) - ZG_E_dstSize_tooSmall
  else result := nBlocks;
end;

end.
