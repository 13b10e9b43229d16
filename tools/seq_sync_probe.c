/* Does a block's sequence bitstream fall into step when decoding starts in the middle?
 *
 * Study tool, CPU only (built and driven by tools/seq_sync_probe.py); it includes the oracle to get every block's
 * tables and bitstream exactly as ZSTD_decodeSeqHeaders leaves them (zstd_decompress_block.pas:690-784) and restates
 * the state half of ZSTD_decodeSequence (:1083-1230) on a logical bit position: per sequence the additional bits of
 * offset, match length and literal length are skipped, then the LL, ML and OF states are updated in that order.
 *
 * For every block the true trajectory is recorded: the bit position and the three states before each sequence.
 * Speculative decoders are then started at evenly spaced places of the stream and run until they stand on a true
 * sequence boundary with all three states equal to the true ones (from there on they would decode the true sequences),
 * or until the stream ends.  Two kinds of start:
 *   mode 0  at a true sequence boundary with arbitrary states (the most a splitter could hope to know)
 *   mode 1  at an arbitrary bit position with arbitrary states
 * Reported: how many starts fell into step at all, and after how many sequences.
 */
#include "../oracle/zstd_oracle.c"

typedef struct {
    /* accumulated over blocks, per mode */
    U64 starts[2], synced[2];
    U64 cursorHits[2];          /* boundaries at which only the bit position coincided with a true boundary */
    U64 steps[2];               /* speculative sequences decoded in all */
    U32* syncAfter[2];          /* steps to fall into step, one entry per synced start */
    size_t nSync[2], capSync[2];
    U64 blocks, sequences;
    U32 K;                      /* starts per block and mode */
    U64 rng;
} Probe;
static Probe g_probe;

static U32 rnd(Probe* p) { p->rng = p->rng * 6364136223846793005ULL + 1442695040888963407ULL; return (U32)(p->rng >> 33); }

/* n <= 32 bits below bit position P of the stream (bit k = bit k&7 of byte k>>3); positions below 0 read as zero */
static U32 bits_below(const BYTE* s, long P, U32 n)
{
    long lo = P - (long)n;
    U64 v;
    if (n == 0) return 0;
    if (lo < 0) {
        if (P <= 0) return 0;
        memcpy(&v, s, 8);
        return (U32)((v & (((U64)1 << P) - 1)) << (-lo));
    }
    memcpy(&v, s + (lo >> 3), 8);
    return (U32)((v >> (lo & 7)) & (((U64)1 << n) - 1));
}

typedef struct { long P; U32 ll, of, ml; } Point;

static void step(const SeqTable* LL, const SeqTable* OF, const SeqTable* ML, const BYTE* s, Point* q)
{
    ZSTD_seqSymbol const l = LL->cells[q->ll], m = ML->cells[q->ml], o = OF->cells[q->of];
    q->P -= (long)o.nbAdditionalBits + m.nbAdditionalBits + l.nbAdditionalBits;
    q->ll = l.nextState + bits_below(s, q->P, l.nbBits); q->P -= l.nbBits;
    q->ml = m.nextState + bits_below(s, q->P, m.nbBits); q->P -= m.nbBits;
    q->of = o.nextState + bits_below(s, q->P, o.nbBits); q->P -= o.nbBits;
}

static void note_sync(Probe* p, int mode, U32 after)
{
    if (p->nSync[mode] == p->capSync[mode]) {
        p->capSync[mode] = p->capSync[mode] ? 2 * p->capSync[mode] : 4096;
        p->syncAfter[mode] = (U32*)realloc(p->syncAfter[mode], p->capSync[mode] * sizeof(U32));
    }
    p->syncAfter[mode][p->nSync[mode]++] = after;
}

static void observe(const SeqTable* LL, const SeqTable* OF, const SeqTable* ML, const BYTE* stream, size_t size, int nbSeq)
{
    Probe* const p = &g_probe;
    BYTE* s;
    long total;
    Point* truth;
    int* at;            /* bit position -> index of the true sequence that starts there, or -1 */
    int i;
    if (size == 0 || stream[size - 1] == 0 || nbSeq < 64) return;
    s = (BYTE*)calloc(size + 16, 1);
    memcpy(s, stream, size);
    total = (long)size * 8 - (long)(8 - highbit32(stream[size - 1]));
    truth = (Point*)malloc(((size_t)nbSeq + 1) * sizeof(Point));
    at = (int*)malloc(((size_t)total + 1) * sizeof(int));
    for (i = 0; i <= total; i++) at[i] = -1;
    {   Point q;
        q.P = total;
        q.ll = bits_below(s, q.P, LL->tableLog); q.P -= LL->tableLog;      /* ZSTD_initFseState: LL, OF, ML (:1266-1268) */
        q.of = bits_below(s, q.P, OF->tableLog); q.P -= OF->tableLog;
        q.ml = bits_below(s, q.P, ML->tableLog); q.P -= ML->tableLog;
        for (i = 0; i < nbSeq; i++) {
            truth[i] = q;
            if (q.P >= 0 && q.P <= total) at[q.P] = i;
            if (i + 1 < nbSeq) step(LL, OF, ML, s, &q);
        }
    }
    if (truth[nbSeq - 1].P < 0) goto done;      /* a damaged stream: not a case for this study */
    p->blocks++; p->sequences += (U64)nbSeq;
    {   U32 k; int mode;
        for (mode = 0; mode < 2; mode++) for (k = 1; k < p->K; k++) {
            int const i0 = (int)((U64)nbSeq * k / p->K);
            Point q;
            U32 n = 0;
            int hit = 0;
            q.P = truth[i0].P - (mode ? (long)(1 + rnd(p) % 24) : 0);
            q.ll = rnd(p) & ((1u << LL->tableLog) - 1);
            q.of = rnd(p) & ((1u << OF->tableLog) - 1);
            q.ml = rnd(p) & ((1u << ML->tableLog) - 1);
            if (mode == 0 && q.ll == truth[i0].ll) q.ll ^= 1u & ((1u << LL->tableLog) - 1);    /* (a start that is the truth proves nothing) */
            p->starts[mode]++;
            while (q.P > 0 && n < (U32)nbSeq) {
                int const t = at[q.P];
                if (t >= 0) {
                    if (truth[t].ll == q.ll && truth[t].of == q.of && truth[t].ml == q.ml && !(mode == 0 && n == 0)) { hit = 1; break; }
                    if (n) p->cursorHits[mode]++;
                }
                step(LL, OF, ML, s, &q);
                n++;
            }
            p->steps[mode] += n;
            if (hit) { p->synced[mode]++; note_sync(p, mode, n); }
        }
    }
done:
    free(at); free(truth); free(s);
}

void probe_begin(unsigned K, unsigned long long seed)
{
    int m;
    for (m = 0; m < 2; m++) free(g_probe.syncAfter[m]);
    memset(&g_probe, 0, sizeof g_probe);
    g_probe.K = K; g_probe.rng = seed;
    g_seqObserver = observe;
}

/* decodes one frame with the observer on; returns the oracle's result */
size_t probe_frame(void* dst, size_t cap, const void* src, size_t n) { return oracle_decompress(dst, cap, src, n); }

static int cmp_u32(const void* a, const void* b) { U32 const x = *(const U32*)a, y = *(const U32*)b; return x < y ? -1 : x > y; }

/* out[mode][...] = {starts, synced, cursorHits, steps, q10, q50, q90, max} ; returns blocks, *sequences total */
unsigned long long probe_end(unsigned long long* out, unsigned long long* sequences)
{
    int m;
    for (m = 0; m < 2; m++) {
        U32* const v = g_probe.syncAfter[m];
        size_t const n = g_probe.nSync[m];
        if (n) qsort(v, n, sizeof(U32), cmp_u32);
        out[8 * m + 0] = g_probe.starts[m]; out[8 * m + 1] = g_probe.synced[m];
        out[8 * m + 2] = g_probe.cursorHits[m]; out[8 * m + 3] = g_probe.steps[m];
        out[8 * m + 4] = n ? v[n / 10] : 0; out[8 * m + 5] = n ? v[n / 2] : 0; out[8 * m + 6] = n ? v[n * 9 / 10] : 0; out[8 * m + 7] = n ? v[n - 1] : 0;
    }
    *sequences = g_probe.sequences;
    g_seqObserver = 0;
    return g_probe.blocks;
}
