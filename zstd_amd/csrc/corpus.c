/*
 * corpus.c -- seeded synthetic corpora for tests and benchmarks (host only, no GPU code).
 *
 * Two generators, both pure integer arithmetic so the same seed gives the same bytes on
 * every machine:
 *   text-like : Zipf-weighted 4000-word a-z vocabulary, 12-word phrases joined by ", " / ".\n"
 *   binary    : packed little-endian 20-byte records
 *               (u32 counter, u16 0..50, u16 in {0,1,2,7}, f64 ~N(100,15) rounded to 0.1, u32 12 random bits)
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>

static uint64_t sm64(uint64_t* s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

#define NWORDS 4000
static uint8_t g_words[NWORDS][12];
static uint8_t g_wlen[NWORDS];
static uint32_t g_cdf[NWORDS];   /* cumulative Zipf weights scaled to 2^32 */
static int g_vocab_ready = 0;

static void build_vocab(void)
{
    uint64_t s = 0x5EEDC0DEULL;
    uint64_t w[NWORDS];
    uint64_t total = 0, acc = 0;
    int i, j;
    for (i = 0; i < NWORDS; i++) {
        int const len = 2 + (int)(sm64(&s) % 9);   /* 2..10 letters */
        g_wlen[i] = (uint8_t)len;
        for (j = 0; j < len; j++) g_words[i][j] = (uint8_t)('a' + sm64(&s) % 26);
        w[i] = (1ULL << 40) / (uint64_t)(i + 1);   /* Zipf weight 1/rank, fixed point */
        total += w[i];
    }
    for (i = 0; i < NWORDS; i++) {
        acc += w[i];
        g_cdf[i] = (uint32_t)(((__uint128_t)acc << 32) / total - (i == NWORDS - 1 ? 1 : 0));
    }
    g_vocab_ready = 1;
}

void corpus_text(uint64_t seed, uint8_t* dst, size_t n)
{
    uint64_t s = seed * 0x9E3779B97F4A7C15ULL + 0x1234567ULL;
    size_t pos = 0;
    unsigned k = 0;
    if (!g_vocab_ready) build_vocab();
    while (pos < n) {
        uint64_t const r = sm64(&s);
        uint32_t const u = (uint32_t)r;
        int lo = 0, hi = NWORDS - 1;
        uint8_t tok[16];
        size_t tl, c;
        while (lo < hi) { int const mid = (lo + hi) >> 1; if (g_cdf[mid] < u) lo = mid + 1; else hi = mid; }
        tl = g_wlen[lo];
        memcpy(tok, g_words[lo], tl);
        if (++k % 12 == 0) {
            if ((r >> 40) & 1) { tok[tl++] = ','; tok[tl++] = ' '; }
            else { tok[tl++] = '.'; tok[tl++] = '\n'; }
        } else tok[tl++] = ' ';
        c = (n - pos < tl) ? n - pos : tl;
        memcpy(dst + pos, tok, c);
        pos += c;
    }
}

void corpus_binary(uint64_t seed, uint8_t* dst, size_t n)
{
    static const uint16_t kEnum[4] = {0, 1, 2, 7};
    uint64_t s = seed * 0xD1342543DE82EF95ULL + 0x7654321ULL;
    uint32_t counter = (uint32_t)(sm64(&s) & 0xFFFFFF);
    size_t pos = 0;
    while (pos < n) {
        uint8_t rec[20];
        uint64_t const a = sm64(&s);
        uint64_t sum = 0;
        int i;
        uint16_t const small = (uint16_t)((a & 0xFFFF) % 51);
        uint16_t const en = kEnum[(a >> 16) & 3];
        uint32_t const bits = (uint32_t)((a >> 20) & 0xFFF);
        int64_t tenths;
        double d;
        size_t c;
        for (i = 0; i < 3; i++) { uint64_t const b = sm64(&s); sum += (b & 0xFFFF) + ((b >> 16) & 0xFFFF) + ((b >> 32) & 0xFFFF) + ((b >> 48) & 0xFFFF); }
        /* Irwin-Hall(12): mean 6, variance 1  ->  100 + 15 z, kept in integer tenths */
        tenths = ((int64_t)1000 * 65536 + (int64_t)150 * ((int64_t)sum - (int64_t)6 * 65536) + 32768) >> 16;
        d = (double)tenths / 10.0;
        memcpy(rec, &counter, 4); counter++;
        memcpy(rec + 4, &small, 2);
        memcpy(rec + 6, &en, 2);
        memcpy(rec + 8, &d, 8);
        memcpy(rec + 16, &bits, 4);
        c = (n - pos < 20) ? n - pos : 20;
        memcpy(dst + pos, rec, c);
        pos += c;
    }
}
