// gen_onehot_avro.cpp -- writes the configs[2]-style synthetic data as RAW input avro for the CLI: `rows` records, 20
// categorical fields x 5000 levels (name = field, term = level, value 1.0), Zipf(1.1) levels, rare positives.
// Build: g++ -O2 -std=c++17 -I ml-ease_amd/host tools/gen_onehot_avro.cpp ml-ease_amd/host/avro_io.o -lz -o tools/gen_onehot_avro
// Use:   tools/gen_onehot_avro <out dir> <rows> [files]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>
#include <sys/stat.h>
#include "avro_io.h"
using namespace mlh;
int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: gen_onehot_avro <out dir> <rows> [files]\n"); return 2; }
    const std::string dir = argv[1];
    const long rows = atol(argv[2]);
    const int files = argc > 3 ? atoi(argv[3]) : 8;
    mkdir(dir.c_str(), 0755);
    const int F = 20, L = 5000;
    std::vector<double> cdf(L);
    double tot = 0;
    for (int i = 0; i < L; i++) tot += std::pow(i + 1.0, -1.1);
    double acc = 0;
    for (int i = 0; i < L; i++) { acc += std::pow(i + 1.0, -1.1) / tot; cdf[i] = acc; }
    std::mt19937_64 rng(20260925);
    std::normal_distribution<double> nd(0, 0.3);
    std::vector<double> beta((size_t)F * L);
    for (auto &b : beta) b = nd(rng);
    std::uniform_real_distribution<double> ud(0, 1);
    const char *schema = "{\"type\":\"record\",\"name\":\"Row\",\"fields\":[{\"name\":\"response\",\"type\":\"int\"},"
                         "{\"name\":\"features\",\"type\":{\"type\":\"array\",\"items\":{\"type\":\"record\",\"name\":\"F\",\"fields\":["
                         "{\"name\":\"name\",\"type\":\"string\"},{\"name\":\"term\",\"type\":\"string\"},{\"name\":\"value\",\"type\":\"float\"}]}}}]}";
    std::vector<std::string> fname(F), lname(L);
    for (int f = 0; f < F; f++) fname[f] = "f" + std::to_string(f);
    for (int l = 0; l < L; l++) lname[l] = std::to_string(l);
    const bool dense = getenv("GEN_DENSE") != nullptr;       // configs[1] style instead: 1000 dense N(0,1) features named "1".."1000"
    const int DF = 1000;
    std::vector<std::string> dname(DF);
    std::vector<double> dbeta(DF);
    std::normal_distribution<double> n01(0, 1), nb(0, 0.1);
    for (int j = 0; j < DF; j++) { dname[j] = std::to_string(j + 1); dbeta[j] = nb(rng); }
    long done = 0;
    for (int fi = 0; fi < files; fi++) {
        char path[512];
        snprintf(path, sizeof path, "%s/part-%05d.avro", dir.c_str(), fi);
        AvroFileWriter w(path, schema, getenv("GEN_CODEC") ? getenv("GEN_CODEC") : "null");
        const long n = rows / files + (fi < rows % files ? 1 : 0);
        int lev[20];
        for (long r = 0; r < n && dense; r++, done++) {
            std::vector<float> x(DF);
            double logit = -1.0;
            for (int j = 0; j < DF; j++) { x[j] = (float)n01(rng); logit += dbeta[j] * x[j]; }
            w.put_long(ud(rng) < 1 / (1 + std::exp(-logit)) ? 1 : 0);
            w.array_start(DF);
            for (int j = 0; j < DF; j++) { w.put_string(dname[j]); w.put_string(""); w.put_float(x[j]); }
            w.array_end();
            w.end_record();
        }
        for (long r = 0; r < n && !dense; r++, done++) {
            double logit = -3.0;
            for (int f = 0; f < F; f++) {
                const double u = ud(rng);
                int lo = 0, hi = L - 1;
                while (lo < hi) { const int mid = (lo + hi) / 2; if (cdf[mid] < u) lo = mid + 1; else hi = mid; }
                lev[f] = lo;
                logit += beta[(size_t)f * L + lo];
            }
            w.put_long(ud(rng) < 1 / (1 + std::exp(-logit)) ? 1 : 0);
            w.array_start(F);
            for (int f = 0; f < F; f++) { w.put_string(fname[f]); w.put_string(lname[lev[f]]); w.put_float(1.0f); }
            w.array_end();
            w.end_record();
        }
        w.close();
    }
    fprintf(stderr, "wrote %ld rows into %d files under %s\n", done, files, dir.c_str());
    return 0;
}
