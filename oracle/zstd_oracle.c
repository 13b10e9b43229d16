/*
 * zstd_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, single-threaded CPU restatement of the one-shot zstd frame
 * decoder of the Free Pascal port terrylao/zstd (itself a transliteration of
 * upstream zstd 1.4.8), and of its bufferless block-by-block interface
 * (ZSTD_decompressContinue, at the end of the file).  It exists to check the HIP decoder; it is never part
 * of the shipped decode path.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.
 *
 * The Pascal source cannot be compiled in this environment and carries porting
 * defects (bitwise `not` for logical not, pointer-variable fills, 32-bit bit
 * container, ...), so every function below follows the *intended* arithmetic
 * of the cited routine.  Parity pin: this file is checked byte-for-byte
 * against libzstd.so.1.4.8 outputs on the fixtures under tests/golden/ (the
 * reference ships no golden vectors of its own).
 *
 * Citations are file:line relative to the reference tree.
 *
 * Malformed input follows the cited routines to the letter wherever they stay
 * inside their buffers: a bitstream that runs dry keeps returning bits of its
 * last container (BIT_lookBits / BIT_lookBitsFast, bitstream.pas:300-314), the
 * double-symbol Huffman decoder is used where HUF_selectDecoder picks it, and
 * the sequence loop is the one ZSTD_decompressBlock_internal of zstd 1.4.8
 * selects (the port's :1541 always takes the prefetching loop; oracle_set_body
 * can force either).  One deliberate difference: a 4-stream Huffman literal
 * section whose 4th segment would start past the end of the literal buffer is
 * rejected (the reference would write outside the buffer).
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>

typedef uint8_t  BYTE;
typedef uint16_t U16;
typedef uint32_t U32;
typedef uint64_t U64;
typedef int16_t  S16;

/* ---- error convention: upstream's (size_t)-code (error_private.pas:6-36 lists the codes;
 *      its ERROR()/ERR_isError at :44-51 are inverted by the port and not reproduced) ---- */
enum {
    ZE_no_error = 0, ZE_GENERIC = 1, ZE_prefix_unknown = 10, ZE_version_unsupported = 12,
    ZE_frameParameter_unsupported = 14, ZE_frameParameter_windowTooLarge = 16,
    ZE_corruption_detected = 20, ZE_checksum_wrong = 22, ZE_dictionary_corrupted = 30,
    ZE_dictionary_wrong = 32, ZE_tableLog_tooLarge = 44, ZE_maxSymbolValue_tooLarge = 46,
    ZE_maxSymbolValue_tooSmall = 48, ZE_dstSize_tooSmall = 70, ZE_srcSize_wrong = 72,
    ZE_maxCode = 120
};
#define ERR(e) ((size_t)0 - (size_t)(ZE_##e))
static int isErr(size_t c) { return c > (size_t)0 - (size_t)ZE_maxCode; }

/* ---- format constants (zstd_internal.pas:9-110, zstd.pas:5-20) ---- */
#define ZSTD_MAGICNUMBER          0xFD2FB528U
#define ZSTD_MAGIC_DICTIONARY     0xEC30A437U
#define ZSTD_MAGIC_SKIPPABLE_START 0x184D2A50U
#define ZSTD_MAGIC_SKIPPABLE_MASK 0xFFFFFFF0U
#define ZSTD_BLOCKSIZE_MAX  (1 << 17)
#define ZSTD_WINDOWLOG_MAX  31
#define ZSTD_CONTENTSIZE_UNKNOWN ((U64)0 - 1)
#define MaxLL 35
#define MaxML 52
#define MaxOff 31
#define LLFSELog 9
#define MLFSELog 9
#define OffFSELog 8
#define LONGNBSEQ 0x7F00
#define HUF_TABLELOG_MAX 12
#define FSE_MIN_TABLELOG 5
#define FSE_TABLELOG_ABSOLUTE_MAX 15
#define MIN_CBLOCK_SIZE 3
#define WILDCOPY_OVERLENGTH 32

/* zstd_ddict.pas:23-49, zstd_internal.pas:57-99 */
static const U32 LL_base[MaxLL + 1] = {
    0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 22, 24, 28, 32, 40,
    48, 64, 0x80, 0x100, 0x200, 0x400, 0x800, 0x1000, 0x2000, 0x4000, 0x8000, 0x10000};
static const U32 LL_bits[MaxLL + 1] = {
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3,
    4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16};
static const U32 ML_base[MaxML + 1] = {
    3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26,
    27, 28, 29, 30, 31, 32, 33, 34, 35, 37, 39, 41, 43, 47, 51, 59, 67, 83, 99, 0x83, 0x103,
    0x203, 0x403, 0x803, 0x1003, 0x2003, 0x4003, 0x8003, 0x10003};
static const U32 ML_bits[MaxML + 1] = {
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
    0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 8, 9, 10, 11,
    12, 13, 14, 15, 16};
static U32 OF_base[MaxOff + 1], OF_bits[MaxOff + 1]; /* OF_base[n] = 2^n - 3 for n>=2, {0,1} first; filled at init */
static const S16 LL_defaultNorm[MaxLL + 1] = {
    4, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2,
    2, 3, 2, 1, 1, 1, 1, 1, -1, -1, -1, -1};
static const S16 ML_defaultNorm[MaxML + 1] = {
    1, 4, 3, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
    1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1,
    -1, -1, -1, -1, -1};
static const S16 OF_defaultNorm[29] = {
    1, 1, 1, 1, 1, 1, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, -1, -1, -1, -1, -1};

/* ---- little-endian reads (zstd.pas:597-716) ---- */
static U32 rdLE16(const BYTE* p) { return (U32)p[0] | ((U32)p[1] << 8); }
static U32 rdLE24(const BYTE* p) { return rdLE16(p) | ((U32)p[2] << 16); }
static U32 rdLE32(const BYTE* p) { return rdLE24(p) | ((U32)p[3] << 24); }
static U64 rdLE64(const BYTE* p) { return (U64)rdLE32(p) | ((U64)rdLE32(p + 4) << 32); }
static U32 highbit32(U32 v) { return 31 - (U32)__builtin_clz(v); } /* bitstream.pas:77-94 */

/* ===================== backward bit reader (bitstream.pas:40-51, :206-398) ===================== */
typedef struct { U64 bitContainer; unsigned bitsConsumed; const BYTE* ptr; const BYTE* start; const BYTE* limitPtr; } BIT_DStream_t;
typedef enum { BIT_DStream_unfinished = 0, BIT_DStream_endOfBuffer = 1, BIT_DStream_completed = 2, BIT_DStream_overflow = 3 } BIT_DStream_status;

/* bitstream.pas:206-275 */
static size_t BIT_initDStream(BIT_DStream_t* bitD, const BYTE* src, size_t srcSize)
{
    if (srcSize < 1) { memset(bitD, 0, sizeof(*bitD)); return ERR(srcSize_wrong); }
    bitD->start = src;
    bitD->limitPtr = bitD->start + sizeof(bitD->bitContainer);
    if (srcSize >= sizeof(bitD->bitContainer)) {
        bitD->ptr = src + srcSize - sizeof(bitD->bitContainer);
        bitD->bitContainer = rdLE64(bitD->ptr);
        { BYTE const lastByte = src[srcSize - 1];
          bitD->bitsConsumed = lastByte ? 8 - highbit32(lastByte) : 0;
          if (lastByte == 0) return ERR(GENERIC); }
    } else {
        bitD->ptr = bitD->start;
        bitD->bitContainer = src[0];
        switch (srcSize) {
        case 7: bitD->bitContainer += (U64)src[6] << 48; /* fall-through */
        case 6: bitD->bitContainer += (U64)src[5] << 40; /* fall-through */
        case 5: bitD->bitContainer += (U64)src[4] << 32; /* fall-through */
        case 4: bitD->bitContainer += (U64)src[3] << 24; /* fall-through */
        case 3: bitD->bitContainer += (U64)src[2] << 16; /* fall-through */
        case 2: bitD->bitContainer += (U64)src[1] << 8;  /* fall-through */
        default: break;
        }
        { BYTE const lastByte = src[srcSize - 1];
          bitD->bitsConsumed = lastByte ? 8 - highbit32(lastByte) : 0;
          if (lastByte == 0) return ERR(corruption_detected); }
        bitD->bitsConsumed += (U32)(sizeof(bitD->bitContainer) - srcSize) * 8;
    }
    return srcSize;
}
/* bitstream.pas:282-337.  Once more bits are asked for than the container holds, both forms keep answering
 * from the container in hand (shift counts wrap at 64): the first takes bits from where the count points after
 * wrapping, the second shifts zeros in from below.  Neither touches memory. */
static U64 BIT_lookBits(const BIT_DStream_t* bitD, U32 nbBits)          /* BIT_getMiddleBits: shift + mask */
{
    U32 const start = 64u - bitD->bitsConsumed - nbBits;                /* wraps like the reference's unsigned arithmetic */
    return (bitD->bitContainer >> (start & 63)) & (((U64)1 << nbBits) - 1);      /* nbBits <= 31 (BIT_mask) */
}
static U64 BIT_lookBitsFast(const BIT_DStream_t* bitD, U32 nbBits)      /* only for nbBits >= 1 */
{
    return (bitD->bitContainer << (bitD->bitsConsumed & 63)) >> ((64 - nbBits) & 63);
}
static void BIT_skipBits(BIT_DStream_t* bitD, U32 nbBits) { bitD->bitsConsumed += nbBits; }
static U64 BIT_readBits(BIT_DStream_t* bitD, U32 nbBits) { U64 const v = BIT_lookBits(bitD, nbBits); BIT_skipBits(bitD, nbBits); return v; }
static U64 BIT_readBitsFast(BIT_DStream_t* bitD, U32 nbBits) { U64 const v = BIT_lookBitsFast(bitD, nbBits); BIT_skipBits(bitD, nbBits); return v; }
/* bitstream.pas:345-354 */
static BIT_DStream_status BIT_reloadDStreamFast(BIT_DStream_t* bitD)
{
    if (bitD->ptr < bitD->limitPtr) return BIT_DStream_overflow;
    bitD->ptr -= bitD->bitsConsumed >> 3; bitD->bitsConsumed &= 7; bitD->bitContainer = rdLE64(bitD->ptr);
    return BIT_DStream_unfinished;
}
/* bitstream.pas:361-390 */
static BIT_DStream_status BIT_reloadDStream(BIT_DStream_t* bitD)
{
    if (bitD->bitsConsumed > 64) return BIT_DStream_overflow;
    if (bitD->ptr >= bitD->limitPtr) {
        bitD->ptr -= bitD->bitsConsumed >> 3; bitD->bitsConsumed &= 7; bitD->bitContainer = rdLE64(bitD->ptr);
        return BIT_DStream_unfinished;
    }
    if (bitD->ptr == bitD->start) {
        if (bitD->bitsConsumed < 64) return BIT_DStream_endOfBuffer;
        return BIT_DStream_completed;
    }
    { U32 nbBytes = bitD->bitsConsumed >> 3;
      BIT_DStream_status result = BIT_DStream_unfinished;
      if (bitD->ptr - nbBytes < bitD->start) { nbBytes = (U32)(bitD->ptr - bitD->start); result = BIT_DStream_endOfBuffer; }
      bitD->ptr -= nbBytes; bitD->bitsConsumed -= nbBytes * 8; bitD->bitContainer = rdLE64(bitD->ptr);
      return result; }
}
/* bitstream.pas:395-398 */
static int BIT_endOfDStream(const BIT_DStream_t* d) { return (d->ptr == d->start) && (d->bitsConsumed == 64); }

/* ===================== FSE_readNCount (entropy_common.pas:83-256) ===================== */
static U32 FSE_ctz(U32 v) { return (U32)__builtin_ctz(v); }
static size_t FSE_readNCount(S16* normalizedCounter, unsigned* maxSVPtr, unsigned* tableLogPtr, const BYTE* headerBuffer, size_t hbSize)
{
    const BYTE* const istart = headerBuffer;
    const BYTE* const iend = istart + hbSize;
    const BYTE* ip = istart;
    int nbBits, remaining, threshold, bitCount;
    U32 bitStream;
    unsigned charnum = 0;
    unsigned const maxSV1 = *maxSVPtr + 1;
    int previous0 = 0;

    if (hbSize < 8) {   /* entropy_common.pas:100-113 : pad short headers into a temp buffer */
        BYTE buffer[8] = {0};
        if (hbSize) memcpy(buffer, headerBuffer, hbSize);
        { size_t const countSize = FSE_readNCount(normalizedCounter, maxSVPtr, tableLogPtr, buffer, sizeof(buffer));
          if (isErr(countSize)) return countSize;
          if (countSize > hbSize) return ERR(corruption_detected);
          return countSize; }
    }
    memset(normalizedCounter, 0, (*maxSVPtr + 1) * sizeof(normalizedCounter[0]));
    bitStream = rdLE32(ip);
    nbBits = (int)(bitStream & 0xF) + FSE_MIN_TABLELOG;
    if (nbBits > FSE_TABLELOG_ABSOLUTE_MAX) return ERR(tableLog_tooLarge);
    bitStream >>= 4; bitCount = 4;
    *tableLogPtr = (unsigned)nbBits;
    remaining = (1 << nbBits) + 1;
    threshold = 1 << nbBits;
    nbBits++;

    for (;;) {
        if (previous0) {
            int repeats = (int)(FSE_ctz(~bitStream | 0x80000000U) >> 1);
            while (repeats >= 12) {
                charnum += 3 * 12;
                if (ip <= iend - 7) { ip += 3; }
                else { bitCount -= (int)(8 * (iend - 7 - ip)); bitCount &= 31; ip = iend - 4; }
                bitStream = rdLE32(ip) >> bitCount;
                repeats = (int)(FSE_ctz(~bitStream | 0x80000000U) >> 1);
            }
            charnum += 3 * (unsigned)repeats;
            bitStream >>= 2 * repeats;
            bitCount += 2 * repeats;
            charnum += bitStream & 3;
            bitCount += 2;
            if (charnum >= maxSV1) break;
            if ((ip <= iend - 7) || (ip + (bitCount >> 3) <= iend - 4)) { ip += bitCount >> 3; bitCount &= 7; }
            else { bitCount -= (int)(8 * (iend - 4 - ip)); bitCount &= 31; ip = iend - 4; }
            bitStream = rdLE32(ip) >> bitCount;
        }
        {   int const max = (2 * threshold - 1) - remaining;
            int count;
            if ((bitStream & (U32)(threshold - 1)) < (U32)max) { count = (int)(bitStream & (U32)(threshold - 1)); bitCount += nbBits - 1; }
            else { count = (int)(bitStream & (U32)(2 * threshold - 1)); if (count >= threshold) count -= max; bitCount += nbBits; }
            count--;
            if (count >= 0) remaining -= count; else remaining += count;
            normalizedCounter[charnum++] = (S16)count;
            previous0 = !count;
            if (remaining < threshold) {
                if (remaining <= 1) break;
                nbBits = (int)highbit32((U32)remaining) + 1;
                threshold = 1 << (nbBits - 1);
            }
            if (charnum >= maxSV1) break;
            if ((ip <= iend - 7) || (ip + (bitCount >> 3) <= iend - 4)) { ip += bitCount >> 3; bitCount &= 7; }
            else { bitCount -= (int)(8 * (iend - 4 - ip)); bitCount &= 31; ip = iend - 4; }
            bitStream = rdLE32(ip) >> bitCount;
        }
    }
    if (remaining != 1) return ERR(corruption_detected);
    if (charnum > maxSV1) return ERR(maxSymbolValue_tooSmall);
    if (bitCount > 32) return ERR(corruption_detected);
    *maxSVPtr = charnum - 1;
    ip += (bitCount + 7) >> 3;
    return (size_t)(ip - istart);
}

/* ===================== FSE byte decoder, used for Huffman weights only ===================== */
typedef struct { U16 newState; BYTE symbol; BYTE nbBits; } FSE_decode_t;   /* fse.pas:57-61 */
#define FSE_TABLESTEP(tableSize) (((tableSize) >> 1) + ((tableSize) >> 3) + 3)  /* fse.pas:92-95 */

/* fse_decompress.pas:40-176 */
static size_t FSE_buildDTable(FSE_decode_t* tableDecode, const S16* normalizedCounter, unsigned maxSymbolValue, unsigned tableLog)
{
    U16 symbolNext[256];
    U32 const maxSV1 = maxSymbolValue + 1;
    U32 const tableSize = 1U << tableLog;
    U32 highThreshold = tableSize - 1;
    U32 s;
    if (maxSymbolValue > 255) return ERR(maxSymbolValue_tooLarge);
    if (tableLog > 12) return ERR(tableLog_tooLarge);
    for (s = 0; s < maxSV1; s++) {
        if (normalizedCounter[s] == -1) { tableDecode[highThreshold--].symbol = (BYTE)s; symbolNext[s] = 1; }
        else symbolNext[s] = (U16)normalizedCounter[s];
    }
    {   U32 const tableMask = tableSize - 1;
        U32 const step = FSE_TABLESTEP(tableSize);
        U32 position = 0;
        for (s = 0; s < maxSV1; s++) {
            int i;
            for (i = 0; i < normalizedCounter[s]; i++) {
                tableDecode[position].symbol = (BYTE)s;
                position = (position + step) & tableMask;
                while (position > highThreshold) position = (position + step) & tableMask;
            }
        }
        if (position != 0) return ERR(GENERIC);
    }
    {   U32 u;
        for (u = 0; u < tableSize; u++) {
            BYTE const symbol = tableDecode[u].symbol;
            U32 const nextState = symbolNext[symbol]++;
            tableDecode[u].nbBits = (BYTE)(tableLog - highbit32(nextState));
            tableDecode[u].newState = (U16)((nextState << tableDecode[u].nbBits) - tableSize);
        }
    }
    return 0;
}

/* fse_decompress.pas:242-316 : two interleaved states; the tail alternates states until the
 * reader overflows.  The 4-symbols-per-iteration fast loop of the original produces the same
 * symbols as this one-at-a-time form because no overflow can happen while >8 bytes remain. */
static size_t FSE_decompress_usingDTable(BYTE* dst, size_t maxDstSize, const BYTE* cSrc, size_t cSrcSize, const FSE_decode_t* dt, unsigned tableLog)
{
    BYTE* op = dst;
    BYTE* const omax = op + maxDstSize;
    BIT_DStream_t bitD;
    size_t state1, state2;
    { size_t const e = BIT_initDStream(&bitD, cSrc, cSrcSize); if (isErr(e)) return e; }
    state1 = (size_t)BIT_readBits(&bitD, tableLog); BIT_reloadDStream(&bitD);
    state2 = (size_t)BIT_readBits(&bitD, tableLog); BIT_reloadDStream(&bitD);
#define FSE_GETSYMBOL(st) do { FSE_decode_t const D = dt[st]; *op++ = D.symbol; st = D.newState + (size_t)BIT_readBits(&bitD, D.nbBits); } while (0)
    for (;;) {
        if (op > (omax - 2)) return ERR(dstSize_tooSmall);
        FSE_GETSYMBOL(state1);
        if (BIT_reloadDStream(&bitD) == BIT_DStream_overflow) { *op++ = dt[state2].symbol; break; }
        if (op > (omax - 2)) return ERR(dstSize_tooSmall);
        FSE_GETSYMBOL(state2);
        if (BIT_reloadDStream(&bitD) == BIT_DStream_overflow) { *op++ = dt[state1].symbol; break; }
    }
#undef FSE_GETSYMBOL
    return (size_t)(op - dst);
}

/* fse_decompress.pas:341-383 (FSE_decompress_wksp_body) */
static size_t FSE_decompress_wksp(BYTE* dst, size_t dstCapacity, const BYTE* cSrc, size_t cSrcSize, unsigned maxLog)
{
    S16 counting[256];
    FSE_decode_t dtable[1 << 6];
    unsigned tableLog, maxSymbolValue = 255;
    size_t const NCountLength = FSE_readNCount(counting, &maxSymbolValue, &tableLog, cSrc, cSrcSize);
    if (isErr(NCountLength)) return NCountLength;
    if (tableLog > maxLog) return ERR(tableLog_tooLarge);
    cSrc += NCountLength; cSrcSize -= NCountLength;
    { size_t const e = FSE_buildDTable(dtable, counting, maxSymbolValue, tableLog); if (isErr(e)) return e; }
    return FSE_decompress_usingDTable(dst, dstCapacity, cSrc, cSrcSize, dtable, tableLog);
}

/* ===================== Huffman (huf_decompress.pas, entropy_common.pas:311-392) ===================== */
static size_t HUF_readStats(BYTE* huffWeight, size_t hwSize, U32* rankStats, U32* nbSymbolsPtr, U32* tableLogPtr, const BYTE* src, size_t srcSize)
{
    U32 weightTotal;
    const BYTE* ip = src;
    size_t iSize, oSize;
    if (!srcSize) return ERR(srcSize_wrong);
    iSize = ip[0];
    if (iSize >= 128) {
        oSize = iSize - 127;
        iSize = (oSize + 1) / 2;
        if (iSize + 1 > srcSize) return ERR(srcSize_wrong);
        if (oSize >= hwSize) return ERR(corruption_detected);
        ip += 1;
        { U32 n; for (n = 0; n < oSize; n += 2) { huffWeight[n] = ip[n / 2] >> 4; huffWeight[n + 1] = ip[n / 2] & 15; } }
    } else {
        if (iSize + 1 > srcSize) return ERR(srcSize_wrong);
        oSize = FSE_decompress_wksp(huffWeight, hwSize - 1, ip + 1, iSize, 6);
        if (isErr(oSize)) return oSize;
    }
    memset(rankStats, 0, (HUF_TABLELOG_MAX + 1) * sizeof(U32));
    weightTotal = 0;
    { U32 n; for (n = 0; n < oSize; n++) {
        if (huffWeight[n] >= HUF_TABLELOG_MAX) return ERR(corruption_detected);
        rankStats[huffWeight[n]]++;
        weightTotal += (1U << huffWeight[n]) >> 1;
    } }
    if (weightTotal == 0) return ERR(corruption_detected);
    { U32 const tableLog = highbit32(weightTotal) + 1;
      if (tableLog > HUF_TABLELOG_MAX) return ERR(corruption_detected);
      *tableLogPtr = tableLog;
      { U32 const total = 1U << tableLog;
        U32 const rest = total - weightTotal;
        U32 const verif = 1U << highbit32(rest);
        U32 const lastWeight = highbit32(rest) + 1;
        if (verif != rest) return ERR(corruption_detected);
        huffWeight[oSize] = (BYTE)lastWeight;
        rankStats[lastWeight]++;
      } }
    if ((rankStats[1] < 2) || (rankStats[1] & 1)) return ERR(corruption_detected);
    *nbSymbolsPtr = (U32)(oSize + 1);
    return iSize + 1;
}

typedef struct { BYTE byte; BYTE nbBits; } HUF_DEltX1;                  /* huf_decompress.pas:41-44 */
typedef struct { U16 sequence; BYTE nbBits; BYTE length; } HUF_DEltX2;  /* huf_decompress.pas:57-61: one or two symbols (first in the low byte) */
typedef struct { BYTE symbol; BYTE weight; } sortedSymbol_t;            /* huf_decompress.pas:62-65 */
/* tableType 0: single-symbol cells, 2^tableLog of them; 1: double-symbol cells, always 2^12 (DTableDesc, huf_decompress.pas:32-37) */
typedef struct { U32 tableLog; U32 tableType; union { HUF_DEltX1 x1[1 << HUF_TABLELOG_MAX]; HUF_DEltX2 x2[1 << HUF_TABLELOG_MAX]; } dt; } HUF_DTable;

/* huf_decompress.pas:133-285 (single-symbol table) */
static size_t HUF_readDTableX1(HUF_DTable* DTable, const BYTE* src, size_t srcSize)
{
    BYTE huffWeight[256];
    U32 rankVal[HUF_TABLELOG_MAX + 1];
    U32 tableLog = 0, nbSymbols = 0;
    size_t const iSize = HUF_readStats(huffWeight, 256, rankVal, &nbSymbols, &tableLog, src, srcSize);
    if (isErr(iSize)) return iSize;
    DTable->tableLog = tableLog; DTable->tableType = 0;
    { U32 n, nextRankStart = 0;
      for (n = 1; n < tableLog + 1; n++) { U32 const current = nextRankStart; nextRankStart += rankVal[n] << (n - 1); rankVal[n] = current; } }
    { U32 n;
      for (n = 0; n < nbSymbols; n++) {
          U32 const w = huffWeight[n];
          U32 const length = (1U << w) >> 1;
          U32 u;
          HUF_DEltX1 D; D.byte = (BYTE)n; D.nbBits = (BYTE)(tableLog + 1 - w);
          for (u = rankVal[w]; u < rankVal[w] + length; u++) DTable->dt.x1[u] = D;
          rankVal[w] += length;
      } }
    return iSize;
}

/* huf_decompress.pas:292-350 : decode one backward stream until the segment is full.  (The reference unrolls by four
 * with a reload per group; a reload per symbol reads the same bits, and a stream that over-consumes fails either way.) */
static size_t HUF_decodeStreamX1(BYTE* p, BYTE* const pEnd, const BYTE* src, size_t srcSize, const HUF_DTable* DTable)
{
    BIT_DStream_t bitD;
    U32 const dtLog = DTable->tableLog;
    { size_t const e = BIT_initDStream(&bitD, src, srcSize); if (isErr(e)) return e; }
    while (p < pEnd) {
        size_t const val = (size_t)BIT_lookBitsFast(&bitD, dtLog);      /* HUF_decodeSymbolX1 :292-299 */
        *p++ = DTable->dt.x1[val].byte;
        BIT_skipBits(&bitD, DTable->dt.x1[val].nbBits);
        BIT_reloadDStream(&bitD);
    }
    BIT_reloadDStream(&bitD);
    if (!BIT_endOfDStream(&bitD)) return ERR(corruption_detected);
    return 0;
}
/* huf_decompress.pas:350-377 */
static size_t HUF_decompress1X1_usingDTable(BYTE* dst, size_t dstSize, const BYTE* cSrc, size_t cSrcSize, const HUF_DTable* DTable)
{
    size_t const e = HUF_decodeStreamX1(dst, dst + dstSize, cSrc, cSrcSize, DTable);
    if (isErr(e)) return e;
    return dstSize;
}
/* huf_decompress.pas:379-559 */
static size_t HUF_decompress4X1_usingDTable(BYTE* dst, size_t dstSize, const BYTE* cSrc, size_t cSrcSize, const HUF_DTable* DTable)
{
    if (cSrcSize < 10) return ERR(corruption_detected);
    {   const BYTE* const istart = cSrc;
        BYTE* const ostart = dst;
        BYTE* const oend = ostart + dstSize;
        size_t const length1 = rdLE16(istart);
        size_t const length2 = rdLE16(istart + 2);
        size_t const length3 = rdLE16(istart + 4);
        size_t const length4 = cSrcSize - (length1 + length2 + length3 + 6);
        const BYTE* const istart1 = istart + 6;
        const BYTE* const istart2 = istart1 + length1;
        const BYTE* const istart3 = istart2 + length2;
        const BYTE* const istart4 = istart3 + length3;
        size_t const segmentSize = (dstSize + 3) / 4;
        BYTE* const opStart2 = ostart + segmentSize;
        BYTE* const opStart3 = opStart2 + segmentSize;
        BYTE* const opStart4 = opStart3 + segmentSize;
        size_t e;
        if (length4 > cSrcSize) return ERR(corruption_detected);
        if (opStart4 > oend) return ERR(corruption_detected);
        e = HUF_decodeStreamX1(ostart,   opStart2, istart1, length1, DTable); if (isErr(e)) return ERR(corruption_detected);
        e = HUF_decodeStreamX1(opStart2, opStart3, istart2, length2, DTable); if (isErr(e)) return ERR(corruption_detected);
        e = HUF_decodeStreamX1(opStart3, opStart4, istart3, length3, DTable); if (isErr(e)) return ERR(corruption_detected);
        e = HUF_decodeStreamX1(opStart4, oend,     istart4, length4, DTable); if (isErr(e)) return ERR(corruption_detected);
        return dstSize;
    }
}

/* ---- double-symbol decoder (huf_decompress.pas:691-1141) ---- */
/* huf_decompress.pas:691-738 */
static void HUF_fillDTableX2Level2(HUF_DEltX2* DTable, U32 sizeLog, const U32 consumed, const U32* rankValOrigin, const int minWeight,
                                   const sortedSymbol_t* sortedSymbols, const U32 sortedListSize, U32 nbBitsBaseline, U16 baseSeq)
{
    HUF_DEltX2 DElt;
    U32 rankVal[HUF_TABLELOG_MAX + 1];
    U32 s;
    memcpy(rankVal, rankValOrigin, sizeof(rankVal));
    if (minWeight > 1) {     /* cells whose second symbol would not fit: the first symbol alone */
        U32 i, skipSize = rankVal[minWeight];
        DElt.sequence = baseSeq; DElt.nbBits = (BYTE)consumed; DElt.length = 1;
        for (i = 0; i < skipSize; i++) DTable[i] = DElt;
    }
    for (s = 0; s < sortedListSize; s++) {
        const U32 symbol = sortedSymbols[s].symbol;
        const U32 weight = sortedSymbols[s].weight;
        const U32 nbBits = nbBitsBaseline - weight;
        const U32 length = 1U << (sizeLog - nbBits);
        const U32 start = rankVal[weight];
        U32 i = start;
        const U32 end = start + length;
        DElt.sequence = (U16)(baseSeq + (symbol << 8)); DElt.nbBits = (BYTE)(nbBits + consumed); DElt.length = 2;
        do { DTable[i++] = DElt; } while (i < end);
        rankVal[weight] += length;
    }
}
typedef U32 rankValCol_t[HUF_TABLELOG_MAX + 1];
/* huf_decompress.pas:741-790 */
static void HUF_fillDTableX2(HUF_DEltX2* DTable, const U32 targetLog, const sortedSymbol_t* sortedList, const U32 sortedListSize,
                             const U32* rankStart, rankValCol_t* rankValOrigin, const U32 maxWeight, const U32 nbBitsBaseline)
{
    U32 rankVal[HUF_TABLELOG_MAX + 1];
    const int scaleLog = (int)nbBitsBaseline - (int)targetLog;
    const U32 minBits = nbBitsBaseline - maxWeight;
    U32 s;
    memcpy(rankVal, rankValOrigin, sizeof(rankVal));
    for (s = 0; s < sortedListSize; s++) {
        const U16 symbol = sortedList[s].symbol;
        const U32 weight = sortedList[s].weight;
        const U32 nbBits = nbBitsBaseline - weight;
        const U32 start = rankVal[weight];
        const U32 length = 1U << (targetLog - nbBits);
        if (targetLog - nbBits >= minBits) {   /* enough room for a second symbol */
            U32 sortedRank;
            int minWeight = (int)nbBits + scaleLog;
            if (minWeight < 1) minWeight = 1;
            sortedRank = rankStart[minWeight];
            HUF_fillDTableX2Level2(DTable + start, targetLog - nbBits, nbBits, rankValOrigin[nbBits], minWeight,
                                   sortedList + sortedRank, sortedListSize - sortedRank, nbBitsBaseline, symbol);
        } else {
            HUF_DEltX2 DElt; U32 u; const U32 end = start + length;
            DElt.sequence = symbol; DElt.nbBits = (BYTE)nbBits; DElt.length = 1;
            for (u = start; u < end; u++) DTable[u] = DElt;
        }
        rankVal[weight] += length;
    }
}
/* huf_decompress.pas:796-910 : the table is always 2^maxTableLog = 2^12 cells */
static size_t HUF_readDTableX2(HUF_DTable* DTable, const BYTE* src, size_t srcSize)
{
    U32 tableLog, maxW, sizeOfSort, nbSymbols;
    U32 const maxTableLog = HUF_TABLELOG_MAX;      /* the decoder context's table: hufTable[0] = HufLog * 0x1000001 (zstd_decompressf.pas:1397) */
    rankValCol_t rankVal[HUF_TABLELOG_MAX];
    U32 rankStats[HUF_TABLELOG_MAX + 1];
    U32 rankStart0[HUF_TABLELOG_MAX + 2];
    U32* const rankStart = rankStart0 + 1;
    sortedSymbol_t sortedSymbol[256];
    BYTE weightList[256];
    size_t iSize;
    memset(rankStats, 0, sizeof(rankStats)); memset(rankStart0, 0, sizeof(rankStart0)); memset(rankVal, 0, sizeof(rankVal));
    iSize = HUF_readStats(weightList, 256, rankStats, &nbSymbols, &tableLog, src, srcSize);
    if (isErr(iSize)) return iSize;
    if (tableLog > maxTableLog) return ERR(tableLog_tooLarge);
    for (maxW = tableLog; rankStats[maxW] == 0; maxW--) {}     /* necessarily finds a solution before 0 */
    {   U32 w, nextRankStart = 0;
        for (w = 1; w < maxW + 1; w++) { U32 curr = nextRankStart; nextRankStart += rankStats[w]; rankStart[w] = curr; }
        rankStart[0] = nextRankStart;      /* put all 0w symbols at the end of sorted list */
        sizeOfSort = nextRankStart;
    }
    {   U32 sidx;
        for (sidx = 0; sidx < nbSymbols; sidx++) {
            U32 const w = weightList[sidx];
            U32 const r = rankStart[w]++;
            sortedSymbol[r].symbol = (BYTE)sidx; sortedSymbol[r].weight = (BYTE)w;
        }
        rankStart[0] = 0;                  /* forget 0w symbols; this is beginning of weight(1) */
    }
    {   U32* const rankVal0 = rankVal[0];
        {   int const rescale = (int)(maxTableLog - tableLog) - 1;   /* tableLog <= maxTableLog */
            U32 nextRankVal = 0, w;
            for (w = 1; w < maxW + 1; w++) { U32 curr = nextRankVal; nextRankVal += rankStats[w] << (w + rescale); rankVal0[w] = curr; }
        }
        {   U32 const minBits = tableLog + 1 - maxW;
            U32 consumed;
            for (consumed = minBits; consumed < maxTableLog - minBits + 1; consumed++) {
                U32* const rankValPtr = rankVal[consumed];
                U32 w;
                for (w = 1; w < maxW + 1; w++) rankValPtr[w] = rankVal0[w] >> consumed;
            }
        }
    }
    HUF_fillDTableX2(DTable->dt.x2, maxTableLog, sortedSymbol, sizeOfSort, rankStart0, rankVal, maxW, tableLog + 1);
    DTable->tableLog = maxTableLog; DTable->tableType = 1;
    return iSize;
}
/* huf_decompress.pas:913-921 */
static U32 HUF_decodeSymbolX2(BYTE* op, BIT_DStream_t* DStream, const HUF_DEltX2* dt, const U32 dtLog)
{
    size_t const val = (size_t)BIT_lookBitsFast(DStream, dtLog);
    op[0] = (BYTE)dt[val].sequence; op[1] = (BYTE)(dt[val].sequence >> 8);       /* always two bytes: the slack is the next segment or the buffer's margin */
    BIT_skipBits(DStream, dt[val].nbBits);
    return dt[val].length;
}
/* huf_decompress.pas:923-942 */
static U32 HUF_decodeLastSymbolX2(BYTE* op, BIT_DStream_t* DStream, const HUF_DEltX2* dt, const U32 dtLog)
{
    size_t const val = (size_t)BIT_lookBitsFast(DStream, dtLog);
    op[0] = (BYTE)dt[val].sequence;
    if (dt[val].length == 1) BIT_skipBits(DStream, dt[val].nbBits);
    else if (DStream->bitsConsumed < 64) {
        BIT_skipBits(DStream, dt[val].nbBits);
        if (DStream->bitsConsumed > 64) DStream->bitsConsumed = 64;    /* the cell's count covers two symbols; only the first was wanted */
    }
    return 1;
}
/* huf_decompress.pas:944-984 */
static size_t HUF_decodeStreamX2(BYTE* p, BIT_DStream_t* bitDPtr, BYTE* const pEnd, const HUF_DEltX2* const dt, const U32 dtLog)
{
    BYTE* const pStart = p;
    while ((BIT_reloadDStream(bitDPtr) == BIT_DStream_unfinished) & ((pEnd - p) > 7)) {      /* p < pEnd - 7 */
        p += HUF_decodeSymbolX2(p, bitDPtr, dt, dtLog);
        p += HUF_decodeSymbolX2(p, bitDPtr, dt, dtLog);
        p += HUF_decodeSymbolX2(p, bitDPtr, dt, dtLog);
        p += HUF_decodeSymbolX2(p, bitDPtr, dt, dtLog);
    }
    while ((BIT_reloadDStream(bitDPtr) == BIT_DStream_unfinished) & ((pEnd - p) >= 2)) p += HUF_decodeSymbolX2(p, bitDPtr, dt, dtLog);
    while ((pEnd - p) >= 2) p += HUF_decodeSymbolX2(p, bitDPtr, dt, dtLog);      /* no need to reload : reached the end of DStream */
    if (p < pEnd) p += HUF_decodeLastSymbolX2(p, bitDPtr, dt, dtLog);
    return (size_t)(p - pStart);
}
/* huf_decompress.pas:986-1014.  dst must have one byte of slack after dstSize (the literal buffer has 32). */
static size_t HUF_decompress1X2_usingDTable(BYTE* dst, size_t dstSize, const BYTE* cSrc, size_t cSrcSize, const HUF_DTable* DTable)
{
    BIT_DStream_t bitD;
    { size_t const e = BIT_initDStream(&bitD, cSrc, cSrcSize); if (isErr(e)) return e; }
    HUF_decodeStreamX2(dst, &bitD, dst + dstSize, DTable->dt.x2, DTable->tableLog);
    if (!BIT_endOfDStream(&bitD)) return ERR(corruption_detected);
    return dstSize;
}
/* huf_decompress.pas:1016-1141 */
static size_t HUF_decompress4X2_usingDTable(BYTE* dst, size_t dstSize, const BYTE* cSrc, size_t cSrcSize, const HUF_DTable* DTable)
{
    if (cSrcSize < 10) return ERR(corruption_detected);
    {   const BYTE* const istart = cSrc;
        BYTE* const ostart = dst;
        BYTE* const oend = ostart + dstSize;
        const HUF_DEltX2* const dt = DTable->dt.x2;
        BIT_DStream_t bitD1, bitD2, bitD3, bitD4;
        size_t const length1 = rdLE16(istart);
        size_t const length2 = rdLE16(istart + 2);
        size_t const length3 = rdLE16(istart + 4);
        size_t const length4 = cSrcSize - (length1 + length2 + length3 + 6);
        const BYTE* const istart1 = istart + 6;
        const BYTE* const istart2 = istart1 + length1;
        const BYTE* const istart3 = istart2 + length2;
        const BYTE* const istart4 = istart3 + length3;
        size_t const segmentSize = (dstSize + 3) / 4;
        BYTE* const opStart2 = ostart + segmentSize;
        BYTE* const opStart3 = opStart2 + segmentSize;
        BYTE* const opStart4 = opStart3 + segmentSize;
        BYTE *op1 = ostart, *op2 = opStart2, *op3 = opStart3, *op4 = opStart4;
        U32 endSignal = 1;
        U32 const dtLog = DTable->tableLog;
        if (length4 > cSrcSize) return ERR(corruption_detected);
        if (opStart4 > oend) return ERR(corruption_detected);      /* (deliberate: see the file header) */
        { size_t const e = BIT_initDStream(&bitD1, istart1, length1); if (isErr(e)) return e; }
        { size_t const e = BIT_initDStream(&bitD2, istart2, length2); if (isErr(e)) return e; }
        { size_t const e = BIT_initDStream(&bitD3, istart3, length3); if (isErr(e)) return e; }
        { size_t const e = BIT_initDStream(&bitD4, istart4, length4); if (isErr(e)) return e; }
        /* 16-32 symbols per loop (4-8 symbols per stream); olimit = oend - 7 */
        while (endSignal && (oend - op4) > 7) {
            int k;
            for (k = 0; k < 4; k++) {
                op1 += HUF_decodeSymbolX2(op1, &bitD1, dt, dtLog);
                op2 += HUF_decodeSymbolX2(op2, &bitD2, dt, dtLog);
                op3 += HUF_decodeSymbolX2(op3, &bitD3, dt, dtLog);
                op4 += HUF_decodeSymbolX2(op4, &bitD4, dt, dtLog);
            }
            endSignal = (U32)((BIT_reloadDStreamFast(&bitD1) == BIT_DStream_unfinished)
                            & (BIT_reloadDStreamFast(&bitD2) == BIT_DStream_unfinished)
                            & (BIT_reloadDStreamFast(&bitD3) == BIT_DStream_unfinished)
                            & (BIT_reloadDStreamFast(&bitD4) == BIT_DStream_unfinished));
        }
        if (op1 > opStart2) return ERR(corruption_detected);
        if (op2 > opStart3) return ERR(corruption_detected);
        if (op3 > opStart4) return ERR(corruption_detected);
        HUF_decodeStreamX2(op1, &bitD1, opStart2, dt, dtLog);
        HUF_decodeStreamX2(op2, &bitD2, opStart3, dt, dtLog);
        HUF_decodeStreamX2(op3, &bitD3, opStart4, dt, dtLog);
        HUF_decodeStreamX2(op4, &bitD4, oend,     dt, dtLog);
        if (!(BIT_endOfDStream(&bitD1) & BIT_endOfDStream(&bitD2) & BIT_endOfDStream(&bitD3) & BIT_endOfDStream(&bitD4))) return ERR(corruption_detected);
        return dstSize;
    }
}
/* HUF_selectDecoder and its cost table (huf_decompress.pas:81-100, :1301-1318): 0 = single-symbol, 1 = double-symbol */
typedef struct { U32 tableTime; U32 decode256Time; } algo_time_t;
static const algo_time_t algoTime[16][3] = {
    {{0,0}, {1,1}, {2,2}},  {{0,0}, {1,1}, {2,2}},
    {{  38,130}, {1313, 74}, {2151, 38}},  {{ 448,128}, {1353, 74}, {2238, 41}},
    {{ 556,128}, {1353, 74}, {2238, 47}},  {{ 714,128}, {1418, 74}, {2436, 53}},
    {{ 883,128}, {1437, 74}, {2464, 61}},  {{ 897,128}, {1515, 75}, {2622, 68}},
    {{ 926,128}, {1613, 75}, {2730, 75}},  {{ 947,128}, {1729, 77}, {3359, 77}},
    {{1107,128}, {2083, 81}, {4006, 84}},  {{1177,128}, {2379, 87}, {4785, 88}},
    {{1242,128}, {2415, 93}, {5155, 84}},  {{1349,128}, {2644,106}, {5260,106}},
    {{1455,128}, {2422,124}, {4174,124}},  {{ 722,128}, {1891,145}, {1936,146}},
};
static int g_hufPolicy = 0;     /* 0: the selector below decides; 1: always the single-symbol decoder (to show where the choice matters) */
void oracle_set_huf(int policy) { g_hufPolicy = policy; }
static U32 HUF_selectDecoder(size_t dstSize, size_t cSrcSize)
{
    U32 const Q = (cSrcSize >= dstSize) ? 15 : (U32)(cSrcSize * 16 / dstSize);
    U32 const D256 = (U32)(dstSize >> 8);
    U32 const DTime0 = algoTime[Q][0].tableTime + (algoTime[Q][0].decode256Time * D256);
    U32 DTime1 = algoTime[Q][1].tableTime + (algoTime[Q][1].decode256Time * D256);
    DTime1 += DTime1 >> 3;      /* advantage to algorithm using less memory, to reduce cache eviction */
    if (g_hufPolicy == 1) return 0;
    return DTime1 < DTime0;
}
/* the decoder the table in hand was built for (HUF_decompress1X_usingDTable_bmi2 / 4X, huf_decompress.pas:1371, :1404) */
static size_t HUF_decompress1X_usingDTable(BYTE* dst, size_t dstSize, const BYTE* cSrc, size_t cSrcSize, const HUF_DTable* DTable)
{
    return DTable->tableType ? HUF_decompress1X2_usingDTable(dst, dstSize, cSrc, cSrcSize, DTable) : HUF_decompress1X1_usingDTable(dst, dstSize, cSrc, cSrcSize, DTable);
}
static size_t HUF_decompress4X_usingDTable(BYTE* dst, size_t dstSize, const BYTE* cSrc, size_t cSrcSize, const HUF_DTable* DTable)
{
    return DTable->tableType ? HUF_decompress4X2_usingDTable(dst, dstSize, cSrc, cSrcSize, DTable) : HUF_decompress4X1_usingDTable(dst, dstSize, cSrc, cSrcSize, DTable);
}

/* ===================== sequence tables (zstd_decompress_block.pas:413-688) ===================== */
typedef struct { U16 nextState; BYTE nbAdditionalBits; BYTE nbBits; U32 baseValue; } ZSTD_seqSymbol;   /* zstd_ddict.pas:64-74 */
typedef struct { U32 tableLog; ZSTD_seqSymbol cells[1 << 9]; } SeqTable;

/* zstd_decompress_block.pas:413-429 */
static void ZSTD_buildSeqTable_rle(SeqTable* dt, U32 baseValue, U32 nbAddBits)
{
    dt->tableLog = 0;
    dt->cells[0].nbBits = 0; dt->cells[0].nextState = 0;
    dt->cells[0].nbAdditionalBits = (BYTE)nbAddBits; dt->cells[0].baseValue = baseValue;
}
/* zstd_decompress_block.pas:436-583.  The port has two spreading loops (with and without
 * low-probability symbols); they lay down the same table, so one loop is kept. */
static void ZSTD_buildFSETable(SeqTable* dt, const S16* normalizedCounter, unsigned maxSymbolValue,
                               const U32* baseValue, const U32* nbAdditionalBits, unsigned tableLog)
{
    ZSTD_seqSymbol* const tableDecode = dt->cells;
    U16 symbolNext[MaxML + 1];
    U32 const maxSV1 = maxSymbolValue + 1;
    U32 const tableSize = 1U << tableLog;
    U32 highThreshold = tableSize - 1;
    U32 s;
    dt->tableLog = tableLog;
    for (s = 0; s < maxSV1; s++) {
        if (normalizedCounter[s] == -1) { tableDecode[highThreshold--].baseValue = s; symbolNext[s] = 1; }
        else symbolNext[s] = (U16)normalizedCounter[s];
    }
    {   U32 const tableMask = tableSize - 1;
        U32 const step = FSE_TABLESTEP(tableSize);
        U32 position = 0;
        for (s = 0; s < maxSV1; s++) {
            int i; int const n = normalizedCounter[s];
            for (i = 0; i < n; i++) {
                tableDecode[position].baseValue = s;
                position = (position + step) & tableMask;
                while (position > highThreshold) position = (position + step) & tableMask;
            }
        }
    }
    {   U32 u;
        for (u = 0; u < tableSize; u++) {
            U32 const symbol = tableDecode[u].baseValue;
            U32 const nextState = symbolNext[symbol]++;
            tableDecode[u].nbBits = (BYTE)(tableLog - highbit32(nextState));
            tableDecode[u].nextState = (U16)((nextState << tableDecode[u].nbBits) - tableSize);
            tableDecode[u].nbAdditionalBits = (BYTE)nbAdditionalBits[symbol];
            tableDecode[u].baseValue = baseValue[symbol];
        }
    }
}

/* ===================== decoder context (zstd_ddict.pas:75-82, :109-170) ===================== */
typedef struct {
    SeqTable LLTable, OFTable, MLTable;          /* entropy.* own storage */
    HUF_DTable hufTable;
    U32 rep[3];
} EntropyTables;

typedef struct oracle_trace_s oracle_trace;

typedef struct {
    EntropyTables entropy;
    const SeqTable *LLTptr, *MLTptr, *OFTptr;
    const HUF_DTable* HUFptr;
    U32 litEntropy, fseEntropy;
    const BYTE *prefixStart, *virtualStart, *dictEnd;
    const BYTE* litPtr; size_t litSize;
    U32 dictID;
    U64 windowSize;            /* fParams.windowSize of the frame being decoded */
    BYTE litBuffer[ZSTD_BLOCKSIZE_MAX + WILDCOPY_OVERLENGTH];
    oracle_trace* trace;
    U32 magicless;             /* ZSTD_d_format = ZSTD_f_zstd1_magicless (zstd.pas:416-417) */
    U32 forceIgnoreChecksum;   /* ZSTD_d_forceIgnoreChecksum (zstd_decompressf.pas:1796-1803) */
} DCtx;

static SeqTable LL_defaultDTable, OF_defaultDTable, ML_defaultDTable;  /* zstd_decompress_block.pas:29-125 */
static int g_init = 0;
static void oracle_init(void)
{
    int n;
    if (g_init) return;
    OF_base[0] = 0; OF_base[1] = 1; OF_bits[0] = 0; OF_bits[1] = 1;
    for (n = 2; n <= MaxOff; n++) { OF_base[n] = (1U << n) - 3; OF_bits[n] = (U32)n; }
    ZSTD_buildFSETable(&LL_defaultDTable, LL_defaultNorm, MaxLL, LL_base, LL_bits, 6);
    ZSTD_buildFSETable(&OF_defaultDTable, OF_defaultNorm, 28, OF_base, OF_bits, 5);
    ZSTD_buildFSETable(&ML_defaultDTable, ML_defaultNorm, MaxML, ML_base, ML_bits, 6);
    g_init = 1;
}

/* ---- optional trace of per-block intermediates, for stage-level checks of the GPU kernels ---- */
struct oracle_trace_s {
    U32 nBlocks, capBlocks;
    U32* blkLitSize;  U32* blkNbSeq;  U32* blkRegen;   /* per compressed block */
    BYTE* lits; size_t litsLen, litsCap;               /* concatenated literal buffers */
    U32* seqs; size_t nSeqs, capSeqs;                   /* triples (litLength, matchLength, offset) */
};
static void trace_block(oracle_trace* t, const BYTE* lit, size_t litSize)
{
    if (!t) return;
    if (t->nBlocks == t->capBlocks) {
        t->capBlocks = t->capBlocks ? t->capBlocks * 2 : 64;
        t->blkLitSize = (U32*)realloc(t->blkLitSize, t->capBlocks * 4);
        t->blkNbSeq = (U32*)realloc(t->blkNbSeq, t->capBlocks * 4);
        t->blkRegen = (U32*)realloc(t->blkRegen, t->capBlocks * 4);
    }
    t->blkLitSize[t->nBlocks] = (U32)litSize; t->blkNbSeq[t->nBlocks] = 0; t->blkRegen[t->nBlocks] = 0;
    t->nBlocks++;
    if (t->litsLen + litSize > t->litsCap) { t->litsCap = (t->litsLen + litSize) * 2 + 4096; t->lits = (BYTE*)realloc(t->lits, t->litsCap); }
    if (litSize) memcpy(t->lits + t->litsLen, lit, litSize);
    t->litsLen += litSize;
}
static void trace_seq(oracle_trace* t, U32 ll, U32 ml, U32 off)
{
    if (!t) return;
    if (t->nSeqs == t->capSeqs) { t->capSeqs = t->capSeqs ? t->capSeqs * 2 : 4096; t->seqs = (U32*)realloc(t->seqs, t->capSeqs * 12); }
    t->seqs[t->nSeqs * 3] = ll; t->seqs[t->nSeqs * 3 + 1] = ml; t->seqs[t->nSeqs * 3 + 2] = off;
    t->nSeqs++; t->blkNbSeq[t->nBlocks - 1]++;
}

/* ===================== literals (zstd_decompress_block.pas:212-411) ===================== */
static size_t ZSTD_decodeLiteralsBlock(DCtx* dctx, const BYTE* src, size_t srcSize)
{
    if (srcSize < MIN_CBLOCK_SIZE) return ERR(corruption_detected);
    {   const BYTE* const istart = src;
        U32 const litEncType = istart[0] & 3;
        switch (litEncType) {
        case 3: /* set_repeat */
            if (dctx->litEntropy == 0) return ERR(dictionary_corrupted);
            /* fall-through */
        case 2: /* set_compressed */
            if (srcSize < 5) return ERR(corruption_detected);
            {   size_t lhSize, litSize, litCSize;
                U32 singleStream = 0;
                U32 const lhlCode = (istart[0] >> 2) & 3;
                U32 const lhc = rdLE32(istart);
                size_t hufSuccess;
                switch (lhlCode) {
                case 0: case 1: default:
                    singleStream = !lhlCode; lhSize = 3; litSize = (lhc >> 4) & 0x3FF; litCSize = (lhc >> 14) & 0x3FF; break;
                case 2: lhSize = 4; litSize = (lhc >> 4) & 0x3FFF; litCSize = lhc >> 18; break;
                case 3: lhSize = 5; litSize = (lhc >> 4) & 0x3FFFF; litCSize = (lhc >> 22) + ((size_t)istart[4] << 10); break;
                }
                if (litSize > ZSTD_BLOCKSIZE_MAX) return ERR(corruption_detected);
                if (litCSize + lhSize > srcSize) return ERR(corruption_detected);
                if (litEncType == 3) {
                    /* huf_decompress.pas:1371 / :1404 : the decoder the stored table was built for */
                    hufSuccess = singleStream
                        ? HUF_decompress1X_usingDTable(dctx->litBuffer, litSize, istart + lhSize, litCSize, dctx->HUFptr)
                        : HUF_decompress4X_usingDTable(dctx->litBuffer, litSize, istart + lhSize, litCSize, dctx->HUFptr);
                } else {
                    /* huf_decompress.pas:1386 (one stream: always the single-symbol decoder) / :1416 (four streams:
                     * HUF_selectDecoder): read the table, then decode */
                    const BYTE* cSrc = istart + lhSize; size_t cSrcSize = litCSize;
                    if (!singleStream && litSize == 0) hufSuccess = ERR(dstSize_tooSmall);
                    else if (!singleStream && cSrcSize == 0) hufSuccess = ERR(corruption_detected);
                    else {
                        U32 const algoNb = singleStream ? 0 : HUF_selectDecoder(litSize, cSrcSize);
                        size_t const hSize = algoNb ? HUF_readDTableX2(&dctx->entropy.hufTable, cSrc, cSrcSize)
                                                    : HUF_readDTableX1(&dctx->entropy.hufTable, cSrc, cSrcSize);
                        if (isErr(hSize)) hufSuccess = hSize;
                        else if (hSize >= cSrcSize) hufSuccess = ERR(srcSize_wrong);
                        else {
                            cSrc += hSize; cSrcSize -= hSize;
                            hufSuccess = singleStream
                                ? HUF_decompress1X1_usingDTable(dctx->litBuffer, litSize, cSrc, cSrcSize, &dctx->entropy.hufTable)
                                : HUF_decompress4X_usingDTable(dctx->litBuffer, litSize, cSrc, cSrcSize, &dctx->entropy.hufTable);
                        }
                    }
                }
                if (isErr(hufSuccess)) return ERR(corruption_detected);
                dctx->litPtr = dctx->litBuffer; dctx->litSize = litSize; dctx->litEntropy = 1;
                if (litEncType == 2) dctx->HUFptr = &dctx->entropy.hufTable;
                memset(dctx->litBuffer + dctx->litSize, 0, WILDCOPY_OVERLENGTH);
                return litCSize + lhSize;
            }
        case 0: /* set_basic */
            {   size_t litSize, lhSize;
                U32 const lhlCode = (istart[0] >> 2) & 3;
                switch (lhlCode) {
                case 0: case 2: default: lhSize = 1; litSize = istart[0] >> 3; break;
                case 1: lhSize = 2; litSize = rdLE16(istart) >> 4; break;
                case 3: lhSize = 3; litSize = rdLE24(istart) >> 4; break;
                }
                if (litSize + lhSize > srcSize) return ERR(corruption_detected);
                dctx->litPtr = istart + lhSize; dctx->litSize = litSize;
                return lhSize + litSize;
            }
        case 1: /* set_rle */
            {   U32 const lhlCode = (istart[0] >> 2) & 3;
                size_t litSize, lhSize;
                switch (lhlCode) {
                case 0: case 2: default: lhSize = 1; litSize = istart[0] >> 3; break;
                case 1: lhSize = 2; litSize = rdLE16(istart) >> 4; break;
                case 3: lhSize = 3; litSize = rdLE24(istart) >> 4;
                        if (srcSize < 4) return ERR(corruption_detected);
                        break;
                }
                if (litSize > ZSTD_BLOCKSIZE_MAX) return ERR(corruption_detected);
                memset(dctx->litBuffer, istart[lhSize], litSize + WILDCOPY_OVERLENGTH);
                dctx->litPtr = dctx->litBuffer; dctx->litSize = litSize;
                return lhSize + 1;
            }
        }
    }
    return ERR(corruption_detected);
}

/* zstd_decompress_block.pas:626-688 */
static size_t ZSTD_buildSeqTable(SeqTable* DTableSpace, const SeqTable** DTablePtr, U32 type, unsigned max, U32 maxLog,
                                 const BYTE* src, size_t srcSize, const U32* baseValue, const U32* nbAdditionalBits,
                                 const SeqTable* defaultTable, U32 flagRepeatTable)
{
    switch (type) {
    case 1: /* set_rle */
        if (!srcSize) return ERR(srcSize_wrong);
        if ((*src) > max) return ERR(corruption_detected);
        { U32 const symbol = *src; ZSTD_buildSeqTable_rle(DTableSpace, baseValue[symbol], nbAdditionalBits[symbol]); }
        *DTablePtr = DTableSpace;
        return 1;
    case 0: /* set_basic */
        *DTablePtr = defaultTable;
        return 0;
    case 3: /* set_repeat */
        if (!flagRepeatTable) return ERR(corruption_detected);
        return 0;
    case 2: /* set_compressed */
        {   unsigned tableLog; S16 norm[MaxML + 1];
            size_t const headerSize = FSE_readNCount(norm, &max, &tableLog, src, srcSize);
            if (isErr(headerSize)) return ERR(corruption_detected);
            if (tableLog > maxLog) return ERR(corruption_detected);
            ZSTD_buildFSETable(DTableSpace, norm, max, baseValue, nbAdditionalBits, tableLog);
            *DTablePtr = DTableSpace;
            return headerSize;
        }
    default: return ERR(GENERIC);
    }
}

/* zstd_decompress_block.pas:690-784 */
static size_t ZSTD_decodeSeqHeaders(DCtx* dctx, int* nbSeqPtr, const BYTE* src, size_t srcSize)
{
    const BYTE* const istart = src;
    const BYTE* const iend = istart + srcSize;
    const BYTE* ip = istart;
    int nbSeq;
    if (srcSize < 1) return ERR(srcSize_wrong);
    nbSeq = *ip++;
    if (!nbSeq) { *nbSeqPtr = 0; if (srcSize != 1) return ERR(srcSize_wrong); return 1; }
    if (nbSeq > 0x7F) {
        if (nbSeq == 0xFF) { if (ip + 2 > iend) return ERR(srcSize_wrong); nbSeq = (int)rdLE16(ip) + LONGNBSEQ; ip += 2; }
        else { if (ip >= iend) return ERR(srcSize_wrong); nbSeq = ((nbSeq - 0x80) << 8) + *ip++; }
    }
    *nbSeqPtr = nbSeq;
    if (ip + 1 > iend) return ERR(srcSize_wrong);
    {   U32 const LLtype = *ip >> 6;
        U32 const OFtype = (*ip >> 4) & 3;
        U32 const MLtype = (*ip >> 2) & 3;
        ip++;
        { size_t const llhSize = ZSTD_buildSeqTable(&dctx->entropy.LLTable, &dctx->LLTptr, LLtype, MaxLL, LLFSELog, ip, (size_t)(iend - ip), LL_base, LL_bits, &LL_defaultDTable, dctx->fseEntropy);
          if (isErr(llhSize)) return ERR(corruption_detected);
          ip += llhSize; }
        { size_t const ofhSize = ZSTD_buildSeqTable(&dctx->entropy.OFTable, &dctx->OFTptr, OFtype, MaxOff, OffFSELog, ip, (size_t)(iend - ip), OF_base, OF_bits, &OF_defaultDTable, dctx->fseEntropy);
          if (isErr(ofhSize)) return ERR(corruption_detected);
          ip += ofhSize; }
        { size_t const mlhSize = ZSTD_buildSeqTable(&dctx->entropy.MLTable, &dctx->MLTptr, MLtype, MaxML, MLFSELog, ip, (size_t)(iend - ip), ML_base, ML_bits, &ML_defaultDTable, dctx->fseEntropy);
          if (isErr(mlhSize)) return ERR(corruption_detected);
          ip += mlhSize; }
    }
    return (size_t)(ip - istart);
}

/* ===================== sequences (zstd_decompress_block.pas:793-1421) ===================== */
typedef struct { size_t litLength, matchLength, offset; } seq_t;
typedef struct { size_t state; const ZSTD_seqSymbol* table; } ZSTD_fseState;
typedef struct { BIT_DStream_t DStream; ZSTD_fseState stateLL, stateOffb, stateML; size_t prevOffset[3]; } seqState_t;

/* zstd_decompress_block.pas:1041-1053 */
static void ZSTD_initFseState(ZSTD_fseState* DStatePtr, BIT_DStream_t* bitD, const SeqTable* dt)
{
    DStatePtr->state = (size_t)BIT_readBits(bitD, dt->tableLog);
    BIT_reloadDStream(bitD);
    DStatePtr->table = dt->cells;
}
/* zstd_decompress_block.pas:1083-1230 (64-bit cadence) */
static seq_t ZSTD_decodeSequence(seqState_t* seqState)
{
    seq_t seq;
    ZSTD_seqSymbol const llDInfo = seqState->stateLL.table[seqState->stateLL.state];
    ZSTD_seqSymbol const mlDInfo = seqState->stateML.table[seqState->stateML.state];
    ZSTD_seqSymbol const ofDInfo = seqState->stateOffb.table[seqState->stateOffb.state];
    U32 const llBase = llDInfo.baseValue, mlBase = mlDInfo.baseValue, ofBase = ofDInfo.baseValue;
    BYTE const llBits = llDInfo.nbAdditionalBits, mlBits = mlDInfo.nbAdditionalBits, ofBits = ofDInfo.nbAdditionalBits;
    BYTE const totalBits = (BYTE)(llBits + mlBits + ofBits);
    {   size_t offset;
        if (ofBits > 1) {
            offset = ofBase + (size_t)BIT_readBitsFast(&seqState->DStream, ofBits);
            seqState->prevOffset[2] = seqState->prevOffset[1];
            seqState->prevOffset[1] = seqState->prevOffset[0];
            seqState->prevOffset[0] = offset;
        } else {
            U32 const ll0 = (llBase == 0);
            if (ofBits == 0) {
                if (!ll0) offset = seqState->prevOffset[0];
                else { offset = seqState->prevOffset[1]; seqState->prevOffset[1] = seqState->prevOffset[0]; seqState->prevOffset[0] = offset; }
            } else {
                offset = ofBase + ll0 + (size_t)BIT_readBitsFast(&seqState->DStream, 1);
                {   size_t temp = (offset == 3) ? seqState->prevOffset[0] - 1 : seqState->prevOffset[offset];
                    temp += !temp;   /* 0 is not valid; input is corrupted; force offset to 1 (:1154, logical not) */
                    if (offset != 1) seqState->prevOffset[2] = seqState->prevOffset[1];
                    seqState->prevOffset[1] = seqState->prevOffset[0];
                    seqState->prevOffset[0] = offset = temp;
                }
            }
        }
        seq.offset = offset;
    }
    seq.matchLength = mlBase;
    if (mlBits > 0) seq.matchLength += (size_t)BIT_readBitsFast(&seqState->DStream, mlBits);
    if (totalBits >= 57 - (LLFSELog + MLFSELog + OffFSELog)) BIT_reloadDStream(&seqState->DStream);   /* :1171-1174 */
    seq.litLength = llBase;
    if (llBits > 0) seq.litLength += (size_t)BIT_readBitsFast(&seqState->DStream, llBits);
    return seq;
}
/* the state update half of ZSTD_decodeSequence (:1220-1225): order LL, ML, OF */
static void ZSTD_updateStates(seqState_t* s)
{
    ZSTD_seqSymbol const ll = s->stateLL.table[s->stateLL.state];
    ZSTD_seqSymbol const ml = s->stateML.table[s->stateML.state];
    ZSTD_seqSymbol const of = s->stateOffb.table[s->stateOffb.state];
    s->stateLL.state = ll.nextState + (size_t)BIT_readBits(&s->DStream, ll.nbBits);
    s->stateML.state = ml.nextState + (size_t)BIT_readBits(&s->DStream, ml.nbBits);
    s->stateOffb.state = of.nextState + (size_t)BIT_readBits(&s->DStream, of.nbBits);
}

/* zstd_decompress_block.pas:896-1039 : copy litLength literals, then matchLength bytes from
 * `offset` back; byte-serial forward copy gives the overlap semantics of ZSTD_overlapCopy8 /
 * ZSTD_safecopy without their over-write slack. */
static size_t ZSTD_execSequence(BYTE* op, BYTE* const oend, seq_t sequence, const BYTE** litPtr, const BYTE* const litLimit,
                                const BYTE* const prefixStart, const BYTE* const virtualStart, const BYTE* const dictEnd)
{
    BYTE* const oLitEnd = op + sequence.litLength;
    size_t const sequenceLength = sequence.litLength + sequence.matchLength;
    const BYTE* match;
    if (sequenceLength > (size_t)(oend - op)) return ERR(dstSize_tooSmall);
    if (sequence.litLength > (size_t)(litLimit - *litPtr)) return ERR(corruption_detected);
    memmove(op, *litPtr, sequence.litLength);
    op = oLitEnd;
    *litPtr += sequence.litLength;
    if (sequence.offset > (size_t)(oLitEnd - prefixStart)) {
        if (sequence.offset > (size_t)(oLitEnd - virtualStart)) return ERR(corruption_detected);
        match = dictEnd - (sequence.offset - (size_t)(oLitEnd - prefixStart));
        if (match + sequence.matchLength <= dictEnd) { memmove(oLitEnd, match, sequence.matchLength); return sequenceLength; }
        {   size_t const length1 = (size_t)(dictEnd - match);
            memmove(oLitEnd, match, length1);
            op = oLitEnd + length1; sequence.matchLength -= length1; match = prefixStart; }
    } else {
        match = oLitEnd - sequence.offset;
    }
    { size_t i; for (i = 0; i < sequence.matchLength; i++) op[i] = match[i]; }
    return sequenceLength;
}

/* zstd_decompress_block.pas:1232-1320 : decode and execute one sequence at a time.  A stream that runs dry is not an
 * error here: the reads keep answering from the last container, and only a stream that is NOT used up after the last
 * sequence is rejected (:1301). */
static size_t ZSTD_decompressSequences(DCtx* dctx, BYTE* dst, size_t maxDstSize, const BYTE* seqStart, size_t seqSize, int nbSeq)
{
    const BYTE* ip = seqStart;
    const BYTE* const iend = ip + seqSize;
    BYTE* const ostart = dst;
    BYTE* const oend = ostart + maxDstSize;
    BYTE* op = ostart;
    const BYTE* litPtr = dctx->litPtr;
    const BYTE* const litEnd = litPtr + dctx->litSize;
    if (nbSeq) {
        seqState_t seqState;
        size_t error = 0;
        dctx->fseEntropy = 1;
        { U32 i; for (i = 0; i < 3; i++) seqState.prevOffset[i] = dctx->entropy.rep[i]; }
        if (isErr(BIT_initDStream(&seqState.DStream, ip, (size_t)(iend - ip)))) return ERR(corruption_detected);
        ZSTD_initFseState(&seqState.stateLL, &seqState.DStream, dctx->LLTptr);
        ZSTD_initFseState(&seqState.stateOffb, &seqState.DStream, dctx->OFTptr);
        ZSTD_initFseState(&seqState.stateML, &seqState.DStream, dctx->MLTptr);
        for (;;) {
            seq_t const sequence = ZSTD_decodeSequence(&seqState);
            ZSTD_updateStates(&seqState);
            trace_seq(dctx->trace, (U32)sequence.litLength, (U32)sequence.matchLength, (U32)sequence.offset);
            {   size_t const oneSeqSize = ZSTD_execSequence(op, oend, sequence, &litPtr, litEnd, dctx->prefixStart, dctx->virtualStart, dctx->dictEnd);
                BIT_reloadDStream(&seqState.DStream);
                if (isErr(oneSeqSize)) { error = oneSeqSize; break; }
                op += oneSeqSize;
                if (--nbSeq == 0) break;
            }
        }
        if (isErr(error)) return error;
        if (nbSeq) return ERR(corruption_detected);
        if (BIT_reloadDStream(&seqState.DStream) < BIT_DStream_completed) return ERR(corruption_detected);
        { U32 i; for (i = 0; i < 3; i++) dctx->entropy.rep[i] = (U32)seqState.prevOffset[i]; }
    }
    {   size_t const lastLLSize = (size_t)(litEnd - litPtr);
        if (lastLLSize > (size_t)(oend - op)) return ERR(dstSize_tooSmall);
        if (lastLLSize) memcpy(op, litPtr, lastLLSize);
        op += lastLLSize;
    }
    return (size_t)(op - ostart);
}

/* zstd_decompress_block.pas:1321-1421 : the loop that decodes four sequences ahead of the one it executes (the
 * prefetches themselves do nothing to the result).  Its verdicts differ from the loop above on malformed input: it stops
 * with corruption_detected as soon as the stream has run dry with sequences left to decode, before it executes the
 * four decoded last, and it does not look at what is left of the stream at the end. */
static size_t ZSTD_decompressSequencesLong(DCtx* dctx, BYTE* dst, size_t maxDstSize, const BYTE* seqStart, size_t seqSize, int nbSeq)
{
#define STORED_SEQS 4
#define STORED_SEQS_MASK (STORED_SEQS - 1)
#define ADVANCED_SEQS 4
    const BYTE* ip = seqStart;
    const BYTE* const iend = ip + seqSize;
    BYTE* const ostart = dst;
    BYTE* const oend = ostart + maxDstSize;
    BYTE* op = ostart;
    const BYTE* litPtr = dctx->litPtr;
    const BYTE* const litEnd = litPtr + dctx->litSize;
    if (nbSeq) {
        seq_t sequences[STORED_SEQS];
        int const seqAdvance = nbSeq < ADVANCED_SEQS ? nbSeq : ADVANCED_SEQS;
        seqState_t seqState;
        int seqNb;
        dctx->fseEntropy = 1;
        { U32 i; for (i = 0; i < 3; i++) seqState.prevOffset[i] = dctx->entropy.rep[i]; }
        if (isErr(BIT_initDStream(&seqState.DStream, ip, (size_t)(iend - ip)))) return ERR(corruption_detected);
        ZSTD_initFseState(&seqState.stateLL, &seqState.DStream, dctx->LLTptr);
        ZSTD_initFseState(&seqState.stateOffb, &seqState.DStream, dctx->OFTptr);
        ZSTD_initFseState(&seqState.stateML, &seqState.DStream, dctx->MLTptr);
        for (seqNb = 0; (BIT_reloadDStream(&seqState.DStream) <= BIT_DStream_completed) && (seqNb < seqAdvance); seqNb++) {
            sequences[seqNb] = ZSTD_decodeSequence(&seqState);
            ZSTD_updateStates(&seqState);
            trace_seq(dctx->trace, (U32)sequences[seqNb].litLength, (U32)sequences[seqNb].matchLength, (U32)sequences[seqNb].offset);
        }
        if (seqNb < seqAdvance) return ERR(corruption_detected);
        for (; (BIT_reloadDStream(&seqState.DStream) <= BIT_DStream_completed) && (seqNb < nbSeq); seqNb++) {
            seq_t const sequence = ZSTD_decodeSequence(&seqState);
            ZSTD_updateStates(&seqState);
            trace_seq(dctx->trace, (U32)sequence.litLength, (U32)sequence.matchLength, (U32)sequence.offset);
            {   size_t const oneSeqSize = ZSTD_execSequence(op, oend, sequences[(seqNb - ADVANCED_SEQS) & STORED_SEQS_MASK], &litPtr, litEnd,
                                                            dctx->prefixStart, dctx->virtualStart, dctx->dictEnd);
                if (isErr(oneSeqSize)) return oneSeqSize;
                sequences[seqNb & STORED_SEQS_MASK] = sequence;
                op += oneSeqSize;
            }
        }
        if (seqNb < nbSeq) return ERR(corruption_detected);
        seqNb -= seqAdvance;
        for (; seqNb < nbSeq; seqNb++) {
            size_t const oneSeqSize = ZSTD_execSequence(op, oend, sequences[seqNb & STORED_SEQS_MASK], &litPtr, litEnd,
                                                        dctx->prefixStart, dctx->virtualStart, dctx->dictEnd);
            if (isErr(oneSeqSize)) return oneSeqSize;
            op += oneSeqSize;
        }
        { U32 i; for (i = 0; i < 3; i++) dctx->entropy.rep[i] = (U32)seqState.prevOffset[i]; }
    }
    {   size_t const lastLLSize = (size_t)(litEnd - litPtr);
        if (lastLLSize > (size_t)(oend - op)) return ERR(dstSize_tooSmall);
        if (lastLLSize) memcpy(op, litPtr, lastLLSize);
        op += lastLLSize;
    }
    return (size_t)(op - ostart);
}

/* zstd_decompress_block.pas:1460-1488 */
static U32 ZSTD_getLongOffsetsShare(const SeqTable* offTable)
{
    U32 const tableLog = offTable->tableLog;
    U32 const max = 1U << tableLog;
    U32 u, total = 0;
    for (u = 0; u < max; u++) if (offTable->cells[u].nbAdditionalBits > 22) total += 1;
    total <<= (OffFSELog - tableLog);      /* scale to OffFSELog */
    return total;
}

/* Which of the two loops runs.  0: as zstd 1.4.8's ZSTD_decompressBlock_internal chooses -- the prefetching loop only
 * for frames with a window above 2^24 whose offset table has a share of long offsets of at least 7/256, and more than
 * 4 sequences (this is the behaviour the libzstd-made fixtures pin); 1: always the prefetching loop, as the port's
 * zstd_decompress_block.pas:1541 is written; 2: never. */
static int g_bodyPolicy = 0;
void oracle_set_body(int policy) { g_bodyPolicy = policy; }

/* Observer of every block's sequence section (the three tables in use, the bitstream, the count), called before the
 * block's sequences are decoded: for study tools that include this file (tools/seq_sync_probe.c); it changes nothing. */
typedef void (*oracle_seq_observer)(const SeqTable* ll, const SeqTable* of, const SeqTable* ml, const BYTE* stream, size_t streamSize, int nbSeq);
static oracle_seq_observer g_seqObserver = 0;

/* zstd_decompress_block.pas:1491-1542 */
static size_t ZSTD_decompressBlock_internal(DCtx* dctx, BYTE* dst, size_t dstCapacity, const BYTE* src, size_t srcSize)
{
    const BYTE* ip = src;
    if (srcSize >= ZSTD_BLOCKSIZE_MAX) return ERR(srcSize_wrong);
    {   size_t const litCSize = ZSTD_decodeLiteralsBlock(dctx, src, srcSize);
        if (isErr(litCSize)) return litCSize;
        ip += litCSize; srcSize -= litCSize;
    }
    trace_block(dctx->trace, dctx->litPtr, dctx->litSize);
    {   int nbSeq;
        size_t const seqHSize = ZSTD_decodeSeqHeaders(dctx, &nbSeq, ip, srcSize);
        size_t r;
        if (isErr(seqHSize)) return seqHSize;
        ip += seqHSize; srcSize -= seqHSize;
        if (g_seqObserver && nbSeq) g_seqObserver(dctx->LLTptr, dctx->OFTptr, dctx->MLTptr, ip, srcSize, nbSeq);
        {   int usePrefetchDecoder = 0;     /* (ddictIsCold concerns digested dictionaries only: not on this path) */
            if (g_bodyPolicy == 1) usePrefetchDecoder = 1;
            else if (g_bodyPolicy == 0 && dctx->windowSize > (1ULL << 24) && nbSeq > ADVANCED_SEQS)
                usePrefetchDecoder = ZSTD_getLongOffsetsShare(dctx->OFTptr) >= 7;
            r = usePrefetchDecoder ? ZSTD_decompressSequencesLong(dctx, dst, dstCapacity, ip, srcSize, nbSeq)
                                   : ZSTD_decompressSequences(dctx, dst, dstCapacity, ip, srcSize, nbSeq);
        }
        if (dctx->trace && !isErr(r)) dctx->trace->blkRegen[dctx->trace->nBlocks - 1] = (U32)r;
        return r;
    }
}

/* ===================== XXH64 (xxhash.pas:18-22, :393-407, :409-488) ===================== */
#define P64_1 0x9E3779B185EBCA87ULL
#define P64_2 0xC2B2AE3D27D4EB4FULL
#define P64_3 0x165667B19E3779F9ULL
#define P64_4 0x85EBCA77C2B2AE63ULL
#define P64_5 0x27D4EB2F165667C5ULL
static U64 rotl64(U64 x, int r) { return (x << r) | (x >> (64 - r)); }
static U64 XXH64_round(U64 acc, U64 input) { acc += input * P64_2; acc = rotl64(acc, 31); acc *= P64_1; return acc; }
static U64 XXH64_mergeRound(U64 acc, U64 val) { val = XXH64_round(0, val); acc ^= val; acc = acc * P64_1 + P64_4; return acc; }
U64 oracle_xxh64(const void* input, size_t len, U64 seed)
{
    const BYTE* p = (const BYTE*)input;
    const BYTE* const bEnd = p + len;
    U64 h64;
    if (len >= 32) {
        const BYTE* const limit = bEnd - 32;
        U64 v1 = seed + P64_1 + P64_2, v2 = seed + P64_2, v3 = seed + 0, v4 = seed - P64_1;
        do {
            v1 = XXH64_round(v1, rdLE64(p)); p += 8;
            v2 = XXH64_round(v2, rdLE64(p)); p += 8;
            v3 = XXH64_round(v3, rdLE64(p)); p += 8;
            v4 = XXH64_round(v4, rdLE64(p)); p += 8;
        } while (p <= limit);
        h64 = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        h64 = XXH64_mergeRound(h64, v1); h64 = XXH64_mergeRound(h64, v2);
        h64 = XXH64_mergeRound(h64, v3); h64 = XXH64_mergeRound(h64, v4);
    } else {
        h64 = seed + P64_5;
    }
    h64 += (U64)len;
    while (p + 8 <= bEnd) { U64 const k1 = XXH64_round(0, rdLE64(p)); h64 ^= k1; h64 = rotl64(h64, 27) * P64_1 + P64_4; p += 8; }
    if (p + 4 <= bEnd) { h64 ^= (U64)rdLE32(p) * P64_1; h64 = rotl64(h64, 23) * P64_2 + P64_3; p += 4; }
    while (p < bEnd) { h64 ^= (*p) * P64_5; h64 = rotl64(h64, 11) * P64_1; p++; }
    h64 ^= h64 >> 33; h64 *= P64_2; h64 ^= h64 >> 29; h64 *= P64_3; h64 ^= h64 >> 32;
    return h64;
}

/* ===================== frame layer (zstd_decompressf.pas) ===================== */
typedef struct { U64 frameContentSize; U64 windowSize; U32 blockSizeMax; U32 dictID; U32 checksumFlag; U32 headerSize; } FrameHeader;
static const size_t ZSTD_fcs_fieldSize[4] = {0, 2, 4, 8};
static const size_t ZSTD_did_fieldSize[4] = {0, 1, 2, 4};

/* zstd_decompressf.pas:246-279; minInputSize = ZSTD_startingInputLength(format) (:97-107): 5, or 1 without the magic number */
static size_t ZSTD_frameHeaderSize(const BYTE* src, size_t srcSize, U32 magicless)
{
    size_t const minInputSize = magicless ? 1 : 5;
    if (srcSize < minInputSize) return ERR(srcSize_wrong);
    {   BYTE const fhd = src[minInputSize - 1];
        U32 const dictID = fhd & 3;
        U32 const singleSegment = (fhd >> 5) & 1;
        U32 const fcsId = fhd >> 6;
        return minInputSize + !singleSegment + ZSTD_did_fieldSize[dictID] + ZSTD_fcs_fieldSize[fcsId] + (singleSegment && !fcsId);
    }
}
/* zstd_decompressf.pas:281-395 ; returns 0 ok, >0 bytes wanted, or error */
static size_t ZSTD_getFrameHeader(FrameHeader* zfh, const BYTE* src, size_t srcSize, U32 magicless)
{
    const BYTE* ip = src;
    size_t const minInputSize = magicless ? 1 : 5;
    memset(zfh, 0, sizeof(*zfh));
    if (srcSize < minInputSize) return minInputSize;
    if (!magicless && rdLE32(src) != ZSTD_MAGICNUMBER) return ERR(prefix_unknown);   /* :301; skippable frames are handled by the caller */
    {   size_t const fhsize = ZSTD_frameHeaderSize(src, srcSize, magicless);
        if (srcSize < fhsize) return fhsize;
        zfh->headerSize = (U32)fhsize;
    }
    {   BYTE const fhdByte = ip[minInputSize - 1];
        size_t pos = minInputSize;
        U32 const dictIDSizeCode = fhdByte & 3;
        U32 const checksumFlag = (fhdByte >> 2) & 1;
        U32 const singleSegment = (fhdByte >> 5) & 1;
        U32 const fcsID = fhdByte >> 6;
        U64 windowSize = 0;
        U32 dictID = 0;
        U64 frameContentSize = ZSTD_CONTENTSIZE_UNKNOWN;
        if ((fhdByte & 0x08) != 0) return ERR(frameParameter_unsupported);
        if (!singleSegment) {
            BYTE const wlByte = ip[pos++];
            U32 const windowLog = (wlByte >> 3) + 10;
            if (windowLog > ZSTD_WINDOWLOG_MAX) return ERR(frameParameter_windowTooLarge);
            windowSize = (1ULL << windowLog);
            windowSize += (windowSize >> 3) * (wlByte & 7);
        }
        switch (dictIDSizeCode) {
        default: case 0: break;
        case 1: dictID = ip[pos]; pos++; break;
        case 2: dictID = rdLE16(ip + pos); pos += 2; break;
        case 3: dictID = rdLE32(ip + pos); pos += 4; break;
        }
        switch (fcsID) {
        default: case 0: if (singleSegment) frameContentSize = ip[pos]; break;
        case 1: frameContentSize = rdLE16(ip + pos) + 256; break;
        case 2: frameContentSize = rdLE32(ip + pos); break;
        case 3: frameContentSize = rdLE64(ip + pos); break;
        }
        if (singleSegment) windowSize = frameContentSize;
        zfh->frameContentSize = frameContentSize;
        zfh->windowSize = windowSize;
        zfh->blockSizeMax = (U32)(windowSize < ZSTD_BLOCKSIZE_MAX ? windowSize : ZSTD_BLOCKSIZE_MAX);
        zfh->dictID = dictID;
        zfh->checksumFlag = checksumFlag;
    }
    return 0;
}

/* zstd_decompressf.pas:1261-1356 */
static size_t ZSTD_loadDEntropy(EntropyTables* entropy, const BYTE* dict, size_t dictSize)
{
    const BYTE* dictPtr = dict;
    const BYTE* const dictEnd = dictPtr + dictSize;
    if (dictSize <= 8) return ERR(dictionary_corrupted);
    dictPtr += 8;
    {   size_t const hSize = g_hufPolicy == 1 ? HUF_readDTableX1(&entropy->hufTable, dictPtr, (size_t)(dictEnd - dictPtr))
                                              : HUF_readDTableX2(&entropy->hufTable, dictPtr, (size_t)(dictEnd - dictPtr));   /* always the double-symbol form (:1288) */
        if (isErr(hSize)) return ERR(dictionary_corrupted);
        dictPtr += hSize;
    }
    {   S16 norm[MaxML + 1]; unsigned maxV = MaxOff, log;
        size_t const h = FSE_readNCount(norm, &maxV, &log, dictPtr, (size_t)(dictEnd - dictPtr));
        if (isErr(h)) return ERR(dictionary_corrupted);
        if (maxV > MaxOff) return ERR(dictionary_corrupted);
        if (log > OffFSELog) return ERR(dictionary_corrupted);
        ZSTD_buildFSETable(&entropy->OFTable, norm, maxV, OF_base, OF_bits, log);
        dictPtr += h;
    }
    {   S16 norm[MaxML + 1]; unsigned maxV = MaxML, log;
        size_t const h = FSE_readNCount(norm, &maxV, &log, dictPtr, (size_t)(dictEnd - dictPtr));
        if (isErr(h)) return ERR(dictionary_corrupted);
        if (maxV > MaxML) return ERR(dictionary_corrupted);
        if (log > MLFSELog) return ERR(dictionary_corrupted);
        ZSTD_buildFSETable(&entropy->MLTable, norm, maxV, ML_base, ML_bits, log);
        dictPtr += h;
    }
    {   S16 norm[MaxML + 1]; unsigned maxV = MaxLL, log;
        size_t const h = FSE_readNCount(norm, &maxV, &log, dictPtr, (size_t)(dictEnd - dictPtr));
        if (isErr(h)) return ERR(dictionary_corrupted);
        if (maxV > MaxLL) return ERR(dictionary_corrupted);
        if (log > LLFSELog) return ERR(dictionary_corrupted);
        ZSTD_buildFSETable(&entropy->LLTable, norm, maxV, LL_base, LL_bits, log);
        dictPtr += h;
    }
    if (dictPtr + 12 > dictEnd) return ERR(dictionary_corrupted);
    {   int i; size_t const dictContentSize = (size_t)(dictEnd - (dictPtr + 12));
        for (i = 0; i < 3; i++) {
            U32 const rep = rdLE32(dictPtr); dictPtr += 4;
            if (rep == 0 || rep > dictContentSize) return ERR(dictionary_corrupted);
            entropy->rep[i] = rep;
        } }
    return (size_t)(dictPtr - dict);
}

/* zstd_decompressf.pas:1386-1408 (ZSTD_decompressBegin) + :1358-1384 (insertDictionary) +
 * :1249-1256 (refDictContent) + zstd_decompress_block.pas:1545-1554 (checkContinuity): in
 * one-shot mode the window of a frame is [dict content][this frame's output]. */
static size_t decompressBegin(DCtx* dctx, const BYTE* dict, size_t dictSize, const BYTE* dst)
{
    dctx->entropy.rep[0] = 1; dctx->entropy.rep[1] = 4; dctx->entropy.rep[2] = 8;   /* zstd_internal.pas:13 */
    dctx->litEntropy = dctx->fseEntropy = 0;
    dctx->dictID = 0;
    dctx->LLTptr = &dctx->entropy.LLTable; dctx->MLTptr = &dctx->entropy.MLTable; dctx->OFTptr = &dctx->entropy.OFTable;
    dctx->HUFptr = &dctx->entropy.hufTable;
    dctx->prefixStart = dst; dctx->virtualStart = dst; dctx->dictEnd = dst;
    if (dict && dictSize) {
        const BYTE* content = dict; size_t contentSize = dictSize;
        if (dictSize >= 8 && rdLE32(dict) == ZSTD_MAGIC_DICTIONARY) {
            dctx->dictID = rdLE32(dict + 4);
            {   size_t const eSize = ZSTD_loadDEntropy(&dctx->entropy, dict, dictSize);
                if (isErr(eSize)) return ERR(dictionary_corrupted);
                content = dict + eSize; contentSize = dictSize - eSize; }
            dctx->litEntropy = dctx->fseEntropy = 1;
        }
        dctx->dictEnd = content + contentSize;
        dctx->virtualStart = dst - contentSize;
    }
    return 0;
}

/* zstd_decompress_block.pas:190-207 */
typedef struct { U32 blockType; U32 lastBlock; U32 origSize; } blockProperties_t;
static size_t ZSTD_getcBlockSize(const BYTE* src, size_t srcSize, blockProperties_t* bp)
{
    if (srcSize < 3) return ERR(srcSize_wrong);
    {   U32 const cBlockHeader = rdLE24(src);
        U32 const cSize = cBlockHeader >> 3;
        bp->lastBlock = cBlockHeader & 1;
        bp->blockType = (cBlockHeader >> 1) & 3;
        bp->origSize = cSize;
        if (bp->blockType == 1) return 1;
        if (bp->blockType == 3) return ERR(corruption_detected);
        return cSize;
    }
}

/* zstd_decompressf.pas:734-834 */
static size_t ZSTD_decompressFrame(DCtx* dctx, BYTE* dst, size_t dstCapacity, const BYTE** srcPtr, size_t* srcSizePtr)
{
    const BYTE* ip = *srcPtr;
    BYTE* const ostart = dst;
    BYTE* const oend = ostart + dstCapacity;
    BYTE* op = ostart;
    size_t remainingSrcSize = *srcSizePtr;
    FrameHeader fh;
    size_t const prefix = dctx->magicless ? 1 : 5;            /* ZSTD_FRAMEHEADERSIZE_PREFIX(format), zstd.pas:615-628 */
    if (remainingSrcSize < prefix + 1 + 3) return ERR(srcSize_wrong);   /* ZSTD_FRAMEHEADERSIZE_MIN(format) + block header (:755) */
    {   size_t const frameHeaderSize = ZSTD_frameHeaderSize(ip, prefix, dctx->magicless);
        if (isErr(frameHeaderSize)) return frameHeaderSize;
        if (remainingSrcSize < frameHeaderSize + 3) return ERR(srcSize_wrong);
        {   size_t const r = ZSTD_getFrameHeader(&fh, ip, frameHeaderSize, dctx->magicless);   /* ZSTD_decodeFrameHeader :541-563 */
            if (isErr(r)) return r;
            if (r > 0) return ERR(srcSize_wrong);
            if (fh.dictID && (dctx->dictID != fh.dictID)) return ERR(dictionary_wrong);
            dctx->windowSize = fh.windowSize;
        }
        ip += frameHeaderSize; remainingSrcSize -= frameHeaderSize;
    }
    for (;;) {
        size_t decodedSize;
        blockProperties_t bp;
        size_t const cBlockSize = ZSTD_getcBlockSize(ip, remainingSrcSize, &bp);
        if (isErr(cBlockSize)) return cBlockSize;
        ip += 3; remainingSrcSize -= 3;
        if (cBlockSize > remainingSrcSize) return ERR(srcSize_wrong);
        switch (bp.blockType) {
        case 2: decodedSize = ZSTD_decompressBlock_internal(dctx, op, (size_t)(oend - op), ip, cBlockSize); break;
        case 0: /* ZSTD_copyRawBlock :699-713 */
            if (cBlockSize > (size_t)(oend - op)) decodedSize = ERR(dstSize_tooSmall);
            else { if (cBlockSize) memcpy(op, ip, cBlockSize); decodedSize = cBlockSize; }
            break;
        case 1: /* ZSTD_setRleBlock :715-731 */
            if (bp.origSize > (size_t)(oend - op)) decodedSize = ERR(dstSize_tooSmall);
            else { if (bp.origSize) memset(op, *ip, bp.origSize); decodedSize = bp.origSize; }
            break;
        default: return ERR(corruption_detected);
        }
        if (isErr(decodedSize)) return decodedSize;
        op += decodedSize;
        ip += cBlockSize; remainingSrcSize -= cBlockSize;
        if (bp.lastBlock) break;
    }
    if (fh.frameContentSize != ZSTD_CONTENTSIZE_UNKNOWN) {
        if ((U64)(op - ostart) != fh.frameContentSize) return ERR(corruption_detected);
    }
    if (fh.checksumFlag) {
        if (remainingSrcSize < 4) return ERR(checksum_wrong);
        if (!dctx->forceIgnoreChecksum) {   /* :819-825 */
            U32 const checkCalc = (U32)oracle_xxh64(ostart, (size_t)(op - ostart), 0);
            U32 const checkRead = rdLE32(ip);
            if (checkRead != checkCalc) return ERR(checksum_wrong);
        }
        ip += 4; remainingSrcSize -= 4;
    }
    *srcPtr = ip; *srcSizePtr = remainingSrcSize;
    return (size_t)(op - ostart);
}

/* zstd_decompressf.pas:836-942 (multi-frame loop), :441-459 (skippable size) */
static size_t decompressMultiFrame(DCtx* dctx, BYTE* dst, size_t dstCapacity, const BYTE* src, size_t srcSize, const BYTE* dict, size_t dictSize)
{
    BYTE* const dststart = dst;
    int moreThan1Frame = 0;
    while (srcSize >= (dctx->magicless ? 1u : 5u)) {    /* ZSTD_startingInputLength(format) (:857) */
        U32 const magicNumber = srcSize >= 4 ? rdLE32(src) : 0;   /* the reference reads 4 bytes whatever is left; fewer cannot hold a frame either way */
        if ((magicNumber & ZSTD_MAGIC_SKIPPABLE_MASK) == ZSTD_MAGIC_SKIPPABLE_START) {
            size_t skippableSize;
            if (srcSize < 8) return ERR(srcSize_wrong);
            {   U32 const sizeU32 = rdLE32(src + 4);
                if ((U32)(sizeU32 + 8) < sizeU32) return ERR(frameParameter_unsupported);
                skippableSize = (size_t)8 + sizeU32;
                if (skippableSize > srcSize) return ERR(srcSize_wrong);
            }
            src += skippableSize; srcSize -= skippableSize;
            continue;
        }
        {   size_t const e = decompressBegin(dctx, dict, dictSize, dst); if (isErr(e)) return e; }
        {   size_t const res = ZSTD_decompressFrame(dctx, dst, dstCapacity, &src, &srcSize);
            if (isErr(res) && ((size_t)0 - res) == ZE_prefix_unknown && moreThan1Frame == 1) return ERR(srcSize_wrong);
            if (isErr(res)) return res;
            dst += res; dstCapacity -= res;
        }
        moreThan1Frame = 1;
    }
    if (srcSize) return ERR(srcSize_wrong);
    return (size_t)(dst - dststart);
}

/* ===================== exported entry points ===================== */
/* ZSTD_decompress / ZSTD_decompressDCtx / ZSTD_decompress_usingDict (zstd_decompressf.pas:971-988, :944-949) */
size_t oracle_decompress_usingDict(void* dst, size_t dstCapacity, const void* src, size_t srcSize, const void* dict, size_t dictSize)
{
    DCtx* const dctx = (DCtx*)calloc(1, sizeof(DCtx));
    size_t r;
    BYTE dummy;
    if (!dctx) return ERR(GENERIC);
    oracle_init();
    r = decompressMultiFrame(dctx, dst ? (BYTE*)dst : &dummy, dst ? dstCapacity : 0, (const BYTE*)src, srcSize, (const BYTE*)dict, dictSize);
    free(dctx);
    return r;
}
/* The same with the decoder parameters of ZSTD_DCtx_setParameter (zstd_decompressf.pas:1763-1807) that change the
 * result of a one-shot call: format (0 zstd1, 1 magicless) and forceIgnoreChecksum. */
size_t oracle_decompress_advanced(void* dst, size_t dstCapacity, const void* src, size_t srcSize, const void* dict, size_t dictSize,
                                  unsigned format, unsigned forceIgnoreChecksum)
{
    DCtx* const dctx = (DCtx*)calloc(1, sizeof(DCtx));
    size_t r;
    BYTE dummy;
    if (!dctx) return ERR(GENERIC);
    oracle_init();
    dctx->magicless = format != 0; dctx->forceIgnoreChecksum = forceIgnoreChecksum != 0;
    r = decompressMultiFrame(dctx, dst ? (BYTE*)dst : &dummy, dst ? dstCapacity : 0, (const BYTE*)src, srcSize, (const BYTE*)dict, dictSize);
    free(dctx);
    return r;
}
size_t oracle_decompress(void* dst, size_t dstCapacity, const void* src, size_t srcSize)
{
    return oracle_decompress_usingDict(dst, dstCapacity, src, srcSize, NULL, 0);
}
unsigned oracle_isError(size_t code) { return (unsigned)isErr(code); }
unsigned oracle_getErrorCode(size_t code) { return isErr(code) ? (unsigned)((size_t)0 - code) : 0; }

/* ZSTD_getFrameContentSize (zstd_decompressf.pas:414-431); upstream sentinels, not the port's 32-bit ones */
U64 oracle_getFrameContentSize(const void* src, size_t srcSize)
{
    FrameHeader zfh;
    /* skippable frame (:303-313 through ZSTD_getFrameHeader): needs its 8-byte header, content size 0 */
    if (srcSize >= 5 && (rdLE32((const BYTE*)src) & ZSTD_MAGIC_SKIPPABLE_MASK) == ZSTD_MAGIC_SKIPPABLE_START)
        return srcSize >= 8 ? 0 : ZSTD_CONTENTSIZE_UNKNOWN - 1;
    if (ZSTD_getFrameHeader(&zfh, (const BYTE*)src, srcSize, 0) != 0) return ZSTD_CONTENTSIZE_UNKNOWN - 1;
    return zfh.frameContentSize;
}
/* ZSTD_findFrameCompressedSize (zstd_decompressf.pas:566-651) */
size_t oracle_findFrameCompressedSize(const void* srcv, size_t srcSize)
{
    const BYTE* const src = (const BYTE*)srcv;
    if (srcSize >= 8 && (rdLE32(src) & ZSTD_MAGIC_SKIPPABLE_MASK) == ZSTD_MAGIC_SKIPPABLE_START) {
        U32 const sizeU32 = rdLE32(src + 4);
        size_t skippableSize;
        if ((U32)(sizeU32 + 8) < sizeU32) return ERR(frameParameter_unsupported);
        skippableSize = (size_t)8 + sizeU32;
        if (skippableSize > srcSize) return ERR(srcSize_wrong);
        return skippableSize;
    }
    if (srcSize >= 5 && (rdLE32(src) & ZSTD_MAGIC_SKIPPABLE_MASK) == ZSTD_MAGIC_SKIPPABLE_START) return ERR(srcSize_wrong);   /* half a skippable header: ZSTD_getFrameHeader wants 8 bytes (:303-308, :599) */
    {   const BYTE* ip = src;
        size_t remainingSize = srcSize;
        FrameHeader zfh;
        {   size_t const ret = ZSTD_getFrameHeader(&zfh, src, srcSize, 0);
            if (isErr(ret)) return ret;
            if (ret > 0) return ERR(srcSize_wrong);
        }
        ip += zfh.headerSize; remainingSize -= zfh.headerSize;
        for (;;) {
            blockProperties_t bp;
            size_t const cBlockSize = ZSTD_getcBlockSize(ip, remainingSize, &bp);
            if (isErr(cBlockSize)) return cBlockSize;
            if (3 + cBlockSize > remainingSize) return ERR(srcSize_wrong);
            ip += 3 + cBlockSize; remainingSize -= 3 + cBlockSize;
            if (bp.lastBlock) break;
        }
        if (zfh.checksumFlag) { if (remainingSize < 4) return ERR(srcSize_wrong); ip += 4; }
        return (size_t)(ip - src);
    }
}

/* ---- the constant tables in use, for the known-answer check against the reference's own copies
 * (zstd_decompress_block.pas:29-125, zstd_ddict.pas:23-49, zstd_internal.pas:57-99).
 * which: 0 LL, 1 OF, 2 ML.  cells: 4 values per cell {nextState, nbAdditionalBits, nbBits, baseValue}; returns the table log.
 * base / bits / norm receive the code tables and the default distribution (MaxLL+1 / MaxOff+1 resp. 29 / MaxML+1 entries). */
unsigned oracle_default_table(int which, U32* cells, U32* base, U32* bits, int* norm)
{
    const SeqTable* t; const U32 *b, *bt; const S16* nm; U32 n, nNorm, u;
    oracle_init();
    if (which == 0) { t = &LL_defaultDTable; b = LL_base; bt = LL_bits; nm = LL_defaultNorm; n = MaxLL + 1; nNorm = MaxLL + 1; }
    else if (which == 1) { t = &OF_defaultDTable; b = OF_base; bt = OF_bits; nm = OF_defaultNorm; n = MaxOff + 1; nNorm = 29; }
    else { t = &ML_defaultDTable; b = ML_base; bt = ML_bits; nm = ML_defaultNorm; n = MaxML + 1; nNorm = MaxML + 1; }
    for (u = 0; u < (1U << t->tableLog); u++) {
        cells[4 * u] = t->cells[u].nextState; cells[4 * u + 1] = t->cells[u].nbAdditionalBits;
        cells[4 * u + 2] = t->cells[u].nbBits; cells[4 * u + 3] = t->cells[u].baseValue;
    }
    for (u = 0; u < n; u++) { base[u] = b[u]; bits[u] = bt[u]; }
    for (u = 0; u < nNorm; u++) norm[u] = nm[u];
    return t->tableLog;
}

/* ---- traced decode: same result as oracle_decompress_usingDict, plus per-block intermediates ---- */
oracle_trace* oracle_trace_new(void) { return (oracle_trace*)calloc(1, sizeof(oracle_trace)); }
void oracle_trace_free(oracle_trace* t) { if (!t) return; free(t->blkLitSize); free(t->blkNbSeq); free(t->blkRegen); free(t->lits); free(t->seqs); free(t); }
U32 oracle_trace_nBlocks(const oracle_trace* t) { return t->nBlocks; }
U64 oracle_trace_nSeqs(const oracle_trace* t) { return t->nSeqs; }
U64 oracle_trace_litBytes(const oracle_trace* t) { return t->litsLen; }
const U32* oracle_trace_blkLitSize(const oracle_trace* t) { return t->blkLitSize; }
const U32* oracle_trace_blkNbSeq(const oracle_trace* t) { return t->blkNbSeq; }
const U32* oracle_trace_blkRegen(const oracle_trace* t) { return t->blkRegen; }
const BYTE* oracle_trace_lits(const oracle_trace* t) { return t->lits; }
const U32* oracle_trace_seqs(const oracle_trace* t) { return t->seqs; }
size_t oracle_decompress_traced(void* dst, size_t dstCapacity, const void* src, size_t srcSize, const void* dict, size_t dictSize, oracle_trace* t)
{
    DCtx* const dctx = (DCtx*)calloc(1, sizeof(DCtx));
    size_t r;
    BYTE dummy;
    if (!dctx) return ERR(GENERIC);
    oracle_init();
    dctx->trace = t;
    r = decompressMultiFrame(dctx, dst ? (BYTE*)dst : &dummy, dst ? dstCapacity : 0, (const BYTE*)src, srcSize, (const BYTE*)dict, dictSize);
    free(dctx);
    return r;
}

/* Batch helper for the CPU baseline timing: decode n independent frames back to back on the
 * calling thread; returns total regenerated bytes or the first error. */
size_t oracle_decompress_batch(BYTE* dstBase, const U64* dstOff, const U64* dstCap,
                               const BYTE* srcBase, const U64* srcOff, const U64* srcSize, U32 n)
{
    size_t total = 0; U32 i;
    for (i = 0; i < n; i++) {
        size_t const r = oracle_decompress(dstBase + dstOff[i], (size_t)dstCap[i], srcBase + srcOff[i], (size_t)srcSize[i]);
        if (isErr(r)) return r;
        total += r;
    }
    return total;
}

/* ===================== bufferless streaming decode (zstd_decompressf.pas:995-1246, :1386-1452) =====================
 * ZSTD_decompressBegin / ZSTD_decompressBegin_usingDict, ZSTD_nextSrcSizeToDecompress, ZSTD_nextInputType and
 * ZSTD_decompressContinue: the caller hands over exactly the bytes the context asks for, one frame header, block header,
 * block or checksum at a time, and every block is decoded straight into the caller's buffer, whose earlier output is the
 * window.  As everywhere in this file the arithmetic is that of zstd 1.4.8, which the port transliterates (the port's
 * ZSTD_decodeFrameHeader, :541-555, lacks the dictionary ID check of 1.4.8; it is kept here). */
typedef enum { ZSTDds_getFrameHeaderSize = 0, ZSTDds_decodeFrameHeader, ZSTDds_decodeBlockHeader, ZSTDds_decompressBlock,
               ZSTDds_decompressLastBlock, ZSTDds_checkChecksum, ZSTDds_decodeSkippableHeader, ZSTDds_skipFrame } ZSTD_dStage;   /* zstd_ddict.pas:96-100 */
/* xxhash.pas:533-545, :686-757, :759-828: the streaming form of XXH64 */
typedef struct { U64 total_len, v1, v2, v3, v4; BYTE mem64[32]; U32 memsize; } XXH64_state_t;
static void XXH64_reset(XXH64_state_t* s, U64 seed)
{
    memset(s, 0, sizeof(*s));
    s->v1 = seed + P64_1 + P64_2; s->v2 = seed + P64_2; s->v3 = seed + 0; s->v4 = seed - P64_1;
}
static void XXH64_update(XXH64_state_t* s, const BYTE* p, size_t len)
{
    const BYTE* const bEnd = p + len;
    s->total_len += len;
    if (s->memsize + len < 32) { if (len) memcpy(s->mem64 + s->memsize, p, len); s->memsize += (U32)len; return; }
    if (s->memsize) {
        memcpy(s->mem64 + s->memsize, p, 32 - s->memsize);
        s->v1 = XXH64_round(s->v1, rdLE64(s->mem64)); s->v2 = XXH64_round(s->v2, rdLE64(s->mem64 + 8));
        s->v3 = XXH64_round(s->v3, rdLE64(s->mem64 + 16)); s->v4 = XXH64_round(s->v4, rdLE64(s->mem64 + 24));
        p += 32 - s->memsize; s->memsize = 0;
    }
    while (p + 32 <= bEnd) {
        s->v1 = XXH64_round(s->v1, rdLE64(p)); s->v2 = XXH64_round(s->v2, rdLE64(p + 8));
        s->v3 = XXH64_round(s->v3, rdLE64(p + 16)); s->v4 = XXH64_round(s->v4, rdLE64(p + 24));
        p += 32;
    }
    if (p < bEnd) { memcpy(s->mem64, p, (size_t)(bEnd - p)); s->memsize = (U32)(bEnd - p); }
}
static U64 XXH64_digest(const XXH64_state_t* s)
{
    const BYTE* p = s->mem64;
    const BYTE* const bEnd = p + s->memsize;
    U64 h64;
    if (s->total_len >= 32) {
        h64 = rotl64(s->v1, 1) + rotl64(s->v2, 7) + rotl64(s->v3, 12) + rotl64(s->v4, 18);
        h64 = XXH64_mergeRound(h64, s->v1); h64 = XXH64_mergeRound(h64, s->v2);
        h64 = XXH64_mergeRound(h64, s->v3); h64 = XXH64_mergeRound(h64, s->v4);
    } else h64 = s->v3 + P64_5;
    h64 += s->total_len;
    while (p + 8 <= bEnd) { U64 const k1 = XXH64_round(0, rdLE64(p)); h64 ^= k1; h64 = rotl64(h64, 27) * P64_1 + P64_4; p += 8; }
    if (p + 4 <= bEnd) { h64 ^= (U64)rdLE32(p) * P64_1; h64 = rotl64(h64, 23) * P64_2 + P64_3; p += 4; }
    while (p < bEnd) { h64 ^= (*p) * P64_5; h64 = rotl64(h64, 11) * P64_1; p++; }
    h64 ^= h64 >> 33; h64 *= P64_2; h64 ^= h64 >> 29; h64 *= P64_3; h64 ^= h64 >> 32;
    return h64;
}

typedef struct {
    DCtx d;
    ZSTD_dStage stage;
    size_t expected, headerSize, rleSize;
    BYTE headerBuffer[18];                  /* ZSTD_FRAMEHEADERSIZE_MAX */
    FrameHeader fParams;
    U32 bType, validateChecksum;
    U64 decodedSize;
    XXH64_state_t xxhState;
    const BYTE* previousDstEnd;
    BYTE* dictCopy;                         /* the dictionary must outlive the frame (ZSTD_refDictContent keeps a pointer): a copy owned here */
} CDCtx;

void* oracle_createDCtx(void) { oracle_init(); return calloc(1, sizeof(CDCtx)); }
void oracle_freeDCtx(void* c) { if (c) { free(((CDCtx*)c)->dictCopy); free(c); } }
/* ZSTD_DCtx_setParameter for the two parameters that reach this path (zstd_decompressf.pas:1763-1807) */
void oracle_DCtx_setFormat(void* c, unsigned magicless, unsigned forceIgnoreChecksum)
{ ((CDCtx*)c)->d.magicless = magicless != 0; ((CDCtx*)c)->d.forceIgnoreChecksum = forceIgnoreChecksum != 0; }

/* zstd_decompressf.pas:1249-1256 */
static void ZSTD_refDictContent(CDCtx* c, const BYTE* dict, size_t dictSize)
{
    c->d.dictEnd = c->previousDstEnd;
    c->d.virtualStart = (const BYTE*)((uintptr_t)dict - ((uintptr_t)c->previousDstEnd - (uintptr_t)c->d.prefixStart));
    c->d.prefixStart = dict;
    c->previousDstEnd = dict + dictSize;
}
/* zstd_decompress_block.pas:1545-1554 */
static void ZSTD_checkContinuity(CDCtx* c, const BYTE* dst)
{
    if (dst != c->previousDstEnd) {   /* not contiguous */
        c->d.dictEnd = c->previousDstEnd;
        c->d.virtualStart = (const BYTE*)((uintptr_t)dst - ((uintptr_t)c->previousDstEnd - (uintptr_t)c->d.prefixStart));
        c->d.prefixStart = dst;
        c->previousDstEnd = dst;
    }
}
/* zstd_decompressf.pas:1386-1408, :1410-1421 (with :1358-1384) */
size_t oracle_decompressBegin_usingDict(void* cv, const void* dict, size_t dictSize)
{
    CDCtx* const c = (CDCtx*)cv;
    DCtx* const dctx = &c->d;
    c->expected = dctx->magicless ? 1 : 5;      /* ZSTD_startingInputLength (:96-107) */
    c->stage = ZSTDds_getFrameHeaderSize;
    c->decodedSize = 0;
    c->previousDstEnd = NULL; dctx->prefixStart = NULL; dctx->virtualStart = NULL; dctx->dictEnd = NULL;
    dctx->litEntropy = dctx->fseEntropy = 0;
    dctx->dictID = 0;
    c->bType = 3;                               /* bt_reserved */
    dctx->entropy.rep[0] = 1; dctx->entropy.rep[1] = 4; dctx->entropy.rep[2] = 8;
    dctx->LLTptr = &dctx->entropy.LLTable; dctx->MLTptr = &dctx->entropy.MLTable; dctx->OFTptr = &dctx->entropy.OFTable;
    dctx->HUFptr = &dctx->entropy.hufTable;
    free(c->dictCopy); c->dictCopy = NULL;
    if (dict && dictSize) {
        const BYTE* content; size_t contentSize = dictSize;
        c->dictCopy = (BYTE*)malloc(dictSize);
        if (!c->dictCopy) return ERR(GENERIC);
        memcpy(c->dictCopy, dict, dictSize);
        content = c->dictCopy;
        if (dictSize >= 8 && rdLE32(content) == ZSTD_MAGIC_DICTIONARY) {
            dctx->dictID = rdLE32(content + 4);
            {   size_t const eSize = ZSTD_loadDEntropy(&dctx->entropy, content, dictSize);
                if (isErr(eSize)) return ERR(dictionary_corrupted);
                content += eSize; contentSize -= eSize; }
            dctx->litEntropy = dctx->fseEntropy = 1;
        }
        ZSTD_refDictContent(c, content, contentSize);
    }
    return 0;
}
size_t oracle_decompressBegin(void* cv) { return oracle_decompressBegin_usingDict(cv, NULL, 0); }
/* zstd_decompressf.pas:995-998 */
size_t oracle_nextSrcSizeToDecompress(void* cv) { return ((CDCtx*)cv)->expected; }
/* zstd_decompressf.pas:1010-1017: a raw block may be handed over in pieces */
static size_t ZSTD_nextSrcSizeToDecompressWithInputSize(const CDCtx* c, size_t inputSize)
{
    if (!(c->stage == ZSTDds_decompressBlock || c->stage == ZSTDds_decompressLastBlock)) return c->expected;
    if (c->bType != 0) return c->expected;
    {   size_t const m = inputSize > 1 ? inputSize : 1;
        return m < c->expected ? m : c->expected; }
}
/* zstd_decompressf.pas:1019-1039: ZSTDnit_frameHeader 0, blockHeader 1, block 2, lastBlock 3, checksum 4, skippableFrame 5 */
unsigned oracle_nextInputType(void* cv)
{
    switch (((CDCtx*)cv)->stage) {
    default:
    case ZSTDds_getFrameHeaderSize: case ZSTDds_decodeFrameHeader: return 0;
    case ZSTDds_decodeBlockHeader: return 1;
    case ZSTDds_decompressBlock: return 2;
    case ZSTDds_decompressLastBlock: return 3;
    case ZSTDds_checkChecksum: return 4;
    case ZSTDds_decodeSkippableHeader: case ZSTDds_skipFrame: return 5;
    }
}
/* zstd_decompressf.pas:1050-1246 */
size_t oracle_decompressContinue(void* cv, void* dstv, size_t dstCapacity, const void* srcv, size_t srcSize)
{
    CDCtx* const c = (CDCtx*)cv;
    DCtx* const dctx = &c->d;
    BYTE* const dst = (BYTE*)dstv;
    const BYTE* const src = (const BYTE*)srcv;
    if (srcSize != ZSTD_nextSrcSizeToDecompressWithInputSize(c, srcSize)) return ERR(srcSize_wrong);
    if (dstCapacity) ZSTD_checkContinuity(c, dst);
    switch (c->stage) {
    case ZSTDds_getFrameHeaderSize:
        if (!dctx->magicless) {
            if (srcSize < 4) return ERR(srcSize_wrong);      /* (the reference reads the magic number whatever it was given) */
            if ((rdLE32(src) & ZSTD_MAGIC_SKIPPABLE_MASK) == ZSTD_MAGIC_SKIPPABLE_START) {
                memcpy(c->headerBuffer, src, srcSize);
                c->expected = 8 - srcSize;                   /* ZSTD_SKIPPABLEHEADERSIZE */
                c->stage = ZSTDds_decodeSkippableHeader;
                return 0;
            }
        }
        c->headerSize = ZSTD_frameHeaderSize(src, srcSize, dctx->magicless);
        if (isErr(c->headerSize)) return c->headerSize;
        memcpy(c->headerBuffer, src, srcSize);
        c->expected = c->headerSize - srcSize;
        c->stage = ZSTDds_decodeFrameHeader;
        return 0;
    case ZSTDds_decodeFrameHeader:
        memcpy(c->headerBuffer + (c->headerSize - srcSize), src, srcSize);
        {   /* ZSTD_decodeFrameHeader (:541-555) */
            size_t const r = ZSTD_getFrameHeader(&c->fParams, c->headerBuffer, c->headerSize, dctx->magicless);
            if (isErr(r)) return r;
            if (r > 0) return ERR(srcSize_wrong);
            if (c->fParams.dictID && (dctx->dictID != c->fParams.dictID)) return ERR(dictionary_wrong);
            c->validateChecksum = (c->fParams.checksumFlag && !dctx->forceIgnoreChecksum) ? 1 : 0;
            if (c->validateChecksum) XXH64_reset(&c->xxhState, 0);
            dctx->windowSize = c->fParams.windowSize;
        }
        c->expected = 3;
        c->stage = ZSTDds_decodeBlockHeader;
        return 0;
    case ZSTDds_decodeBlockHeader:
        {   blockProperties_t bp;
            size_t const cBlockSize = ZSTD_getcBlockSize(src, 3, &bp);
            if (isErr(cBlockSize)) return cBlockSize;
            if (cBlockSize > c->fParams.blockSizeMax) return ERR(corruption_detected);
            c->expected = cBlockSize;
            c->bType = bp.blockType;
            c->rleSize = bp.origSize;
            if (cBlockSize) { c->stage = bp.lastBlock ? ZSTDds_decompressLastBlock : ZSTDds_decompressBlock; return 0; }
            if (bp.lastBlock) {      /* empty block */
                if (c->fParams.checksumFlag) { c->expected = 4; c->stage = ZSTDds_checkChecksum; }
                else { c->expected = 0; c->stage = ZSTDds_getFrameHeaderSize; }
            } else { c->expected = 3; c->stage = ZSTDds_decodeBlockHeader; }
            return 0;
        }
    case ZSTDds_decompressLastBlock:
    case ZSTDds_decompressBlock:
        {   size_t rSize;
            switch (c->bType) {
            case 2: rSize = ZSTD_decompressBlock_internal(dctx, dst, dstCapacity, src, srcSize); c->expected = 0; break;
            case 0:
                if (srcSize > dstCapacity) return ERR(dstSize_tooSmall);      /* ZSTD_copyRawBlock (:699-713) */
                if (srcSize) memcpy(dst, src, srcSize);
                rSize = srcSize;
                c->expected -= rSize;
                break;
            case 1:
                if (c->rleSize > dstCapacity) rSize = ERR(dstSize_tooSmall);   /* ZSTD_setRleBlock (:715-731) */
                else { if (c->rleSize) memset(dst, *src, c->rleSize); rSize = c->rleSize; }
                c->expected = 0;
                break;
            default: return ERR(corruption_detected);
            }
            if (isErr(rSize)) return rSize;
            if (rSize > c->fParams.blockSizeMax) return ERR(corruption_detected);
            c->decodedSize += rSize;
            if (c->validateChecksum) XXH64_update(&c->xxhState, dst, rSize);
            c->previousDstEnd = dst + rSize;
            if (c->expected > 0) return rSize;      /* the rest of a raw block is still to come */
            if (c->stage == ZSTDds_decompressLastBlock) {
                if (c->fParams.frameContentSize != ZSTD_CONTENTSIZE_UNKNOWN && c->decodedSize != c->fParams.frameContentSize) return ERR(corruption_detected);
                if (c->fParams.checksumFlag) { c->expected = 4; c->stage = ZSTDds_checkChecksum; }
                else { c->expected = 0; c->stage = ZSTDds_getFrameHeaderSize; }
            } else { c->stage = ZSTDds_decodeBlockHeader; c->expected = 3; }
            return rSize;
        }
    case ZSTDds_checkChecksum:
        if (c->validateChecksum) {
            U32 const h32 = (U32)XXH64_digest(&c->xxhState);
            if (rdLE32(src) != h32) return ERR(checksum_wrong);
        }
        c->expected = 0;
        c->stage = ZSTDds_getFrameHeaderSize;
        return 0;
    case ZSTDds_decodeSkippableHeader:
        memcpy(c->headerBuffer + (8 - srcSize), src, srcSize);
        c->expected = rdLE32(c->headerBuffer + 4);
        c->stage = ZSTDds_skipFrame;
        return 0;
    case ZSTDds_skipFrame:
        c->expected = 0;
        c->stage = ZSTDds_getFrameHeaderSize;
        return 0;
    default:
        return ERR(GENERIC);
    }
}
