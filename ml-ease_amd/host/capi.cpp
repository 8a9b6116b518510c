// capi.cpp -- C view of the native host pieces (libmlease_host.so) so that the CPU test-suite can check them
// against the Python mirror (ml-ease_amd/dataset.py, admm.py) without a GPU. Not part of the drop-in boundary.
#include <cstring>
#include <string>

#include "avro_io.h"
#include "dataset_builder.h"
#include "java_compat.h"

using namespace mlh;

namespace {
thread_local std::string g_err;
}

extern "C" {

const char *mlh_last_error(void) { return g_err.c_str(); }

// Index an input path. prepared != 0: rows are RegressionPrepareOutput records. Returns NULL on error.
void *mlh_build(const char *path, int num_blocks, const char *map_key, int binary, int num_click_replicates,
                unsigned long long seed, int prepared, int short_index)
{
    try {
        PrepareOptions po;
        po.num_blocks = num_blocks;
        po.map_key = map_key ? map_key : "";
        po.binary_feature = binary != 0;
        po.num_click_replicates = num_click_replicates;
        po.seed = seed;
        po.short_feature_index = short_index != 0;
        DatasetBuilder b(po);
        if (prepared) read_input_rows(path, "key", true, [&](InputRow &r) { b.add_prepared(r); }, b.interner());
        else read_input_rows(path, po.map_key, !po.binary_feature, [&](InputRow &r) { b.add_raw(r); }, b.interner());
        return new Dataset(b.finish());
    } catch (const std::exception &e) {
        g_err = e.what();
        return nullptr;
    }
}
void mlh_free(void *d) { delete static_cast<Dataset *>(d); }
int mlh_n_global(void *d) { return static_cast<Dataset *>(d)->n_global(); }
const char *mlh_feature_name(void *d, int j) { return static_cast<Dataset *>(d)->names[(size_t)j].c_str(); }
int mlh_part_sizes(void *d, int k, long long *out3)
{
    auto &p = static_cast<Dataset *>(d)->parts[(size_t)k];
    out3[0] = p.rows(); out3[1] = p.n_local(); out3[2] = (long long)p.col.size();
    return 0;
}
const long long *mlh_part_rowptr(void *d, int k) { return (const long long *)static_cast<Dataset *>(d)->parts[(size_t)k].row_ptr.data(); }
const int *mlh_part_col(void *d, int k) { return static_cast<Dataset *>(d)->parts[(size_t)k].col.data(); }
const float *mlh_part_val(void *d, int k) { auto &v = static_cast<Dataset *>(d)->parts[(size_t)k].val; return v.empty() ? nullptr : v.data(); }
const signed char *mlh_part_y(void *d, int k) { return (const signed char *)static_cast<Dataset *>(d)->parts[(size_t)k].y.data(); }
const float *mlh_part_weight(void *d, int k) { return static_cast<Dataset *>(d)->parts[(size_t)k].weight.data(); }
const float *mlh_part_offset(void *d, int k) { return static_cast<Dataset *>(d)->parts[(size_t)k].offset.data(); }
const int *mlh_part_l2g(void *d, int k) { return static_cast<Dataset *>(d)->parts[(size_t)k].l2g.data(); }

// Test rows: returns a heap object; sizes via mlh_test_sizes, arrays via the accessors.
void *mlh_test_rows(const char *first_file, void *d, int binary, long long max_rows)
{
    try {
        return new TestRowsData(build_test_rows(first_file, *static_cast<Dataset *>(d), binary != 0, max_rows));
    } catch (const std::exception &e) {
        g_err = e.what();
        return nullptr;
    }
}
void mlh_test_free(void *t) { delete static_cast<TestRowsData *>(t); }
double mlh_test_sizes(void *t, long long *out2)
{
    auto *x = static_cast<TestRowsData *>(t);
    out2[0] = (long long)x->response.size(); out2[1] = (long long)x->gidx.size();
    return x->n;
}
const long long *mlh_test_rowptr(void *t) { return (const long long *)static_cast<TestRowsData *>(t)->row_ptr.data(); }
const int *mlh_test_gidx(void *t) { return static_cast<TestRowsData *>(t)->gidx.data(); }
const double *mlh_test_val(void *t) { auto &v = static_cast<TestRowsData *>(t)->val; return v.empty() ? nullptr : v.data(); }
const signed char *mlh_test_response(void *t) { return (const signed char *)static_cast<TestRowsData *>(t)->response.data(); }
const double *mlh_test_weight(void *t) { return static_cast<TestRowsData *>(t)->weight.data(); }
const double *mlh_test_offset(void *t) { return static_cast<TestRowsData *>(t)->offset.data(); }

int mlh_float_to_string(float f, char *buf, int len)
{
    std::string s = java_float_to_string(f);
    strncpy(buf, s.c_str(), (size_t)len - 1);
    buf[len - 1] = 0;
    return (int)s.size();
}
double mlh_float_string_roundtrip(float f) { return float_string_roundtrip(f); }

}  // extern "C"
