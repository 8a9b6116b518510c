/* synth.c -- C generator of the dense benchmark data (BASELINE configs[1]); TEST INFRASTRUCTURE like the rest of oracle/.
 * Same integer construction as tools/synth_data.py (NumPy) and its torch twin: element (r, c) is an Irwin-Hall(12)
 * variate from twelve 16-bit chunks of three 64-bit counter hashes, (sum - 6*65535) / 65536, so all three generators
 * give identical float32 values; used by tests/golden/make_ref_loglik.py to build the full 1M x 1K job on the CPU quickly. */
#include <math.h>
#include <stdint.h>

static inline uint64_t mix64(uint64_t x)
{
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

static inline uint64_t hash_ctr(uint64_t counter, uint64_t key) { return mix64((counter + 1) * 0x9E3779B97F4A7C15ULL + key); }

static uint64_t stream_key(uint64_t seed, uint64_t stream, uint64_t k)
{
    return seed * 0xD1342543DE82EF95ULL + stream * 0x2545F4914F6CDD1DULL + (k + 1) * 0x9E6C63D0676A9A99ULL;
}

static inline int64_t ih12(uint64_t counter, const uint64_t key[3])
{
    int64_t s = 0;
    for (int k = 0; k < 3; k++) {
        const uint64_t h = hash_ctr(counter, key[k]);
        s += (int64_t)(h & 0xFFFF) + (int64_t)((h >> 16) & 0xFFFF) + (int64_t)((h >> 32) & 0xFFFF) + (int64_t)(h >> 48);
    }
    return s - 6 * 65535;
}

/* rows row0, row0+stride, ... (`rows` of them): X[rows][nfeat] float32, y[rows] +1/-1; beta[nfeat] as tools/synth_data.py dense_beta() */
void orc_synth_dense(uint64_t seed, uint64_t stream, int64_t row0, int64_t stride, int rows, int nfeat, double bias, const double *beta,
                     float *X, int8_t *y)
{
    uint64_t key[3];
    for (int k = 0; k < 3; k++) key[k] = stream_key(seed, stream, (uint64_t)k);
    const uint64_t ukey = stream_key(seed, stream + 100, 7);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < rows; i++) {
        const uint64_t r = (uint64_t)(row0 + stride * (int64_t)i);
        float *xr = X + (int64_t)i * nfeat;
        double logit = 0.0;
        for (int c = 0; c < nfeat; c++) {
            const float x = (float)ih12(r * (uint64_t)nfeat + (uint64_t)c, key) / 65536.0f;
            xr[c] = x;
            logit += (double)x * beta[c];
        }
        logit += bias;
        const double u = (double)(hash_ctr(r, ukey) >> 11) * (1.0 / 9007199254740992.0);
        y[i] = (u < 1.0 / (1.0 + exp(-logit))) ? 1 : -1;
    }
}
