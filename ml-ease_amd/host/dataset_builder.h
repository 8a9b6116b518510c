// dataset_builder.h -- native row preparation + partition indexing.
//
// Mirrors  RegressionPrepare.RegressionPrepareMapper.map      jobs/RegressionPrepare.java:96-191
//          AdmmPartitioner.getPartition                       jobs/RegressionAdmmTrain.java:579-590
//          LibLinearDataset.addInstanceAvro + finish          liblinearfunc/LibLinearDataset.java:413-484,586-658
//          LibLinearBinaryDataset.addInstanceAvro             liblinearfunc/LibLinearBinaryDataset.java:426-515
// and produces, per partition, exactly the arrays mlx_add_partition_csr takes (include/mlease_admm.h).
#pragma once
#include <cstdint>
#include <cstring>
#include <functional>
#include <random>
#include <string>
#include <unordered_map>
#include <vector>

#include "avro_io.h"

namespace mlh {

// name key (name [+ U+0001 + term]) -> dense id in first-seen order. Open addressing over one byte arena: no allocation
// and no std::string per looked-up key (the hot loop of the loader: one lookup per non-zero of the input).
class KeyInterner {
  public:
    KeyInterner() { table_.assign(1u << 16, -1); }
    int32_t intern(const char *p, size_t n)
    {
        const uint64_t h = hash(p, n);
        size_t mask = table_.size() - 1, i = (size_t)h & mask;
        for (;;) {
            const int32_t id = table_[i];
            if (id < 0) break;
            if (hashes_[(size_t)id] == h && len_[(size_t)id] == n && memcmp(arena_.data() + off_[(size_t)id], p, n) == 0) return id;
            i = (i + 1) & mask;
        }
        const int32_t id = (int32_t)off_.size();
        off_.push_back(arena_.size()); len_.push_back((uint32_t)n); hashes_.push_back(h);
        arena_.insert(arena_.end(), p, p + n);
        table_[i] = id;
        if ((size_t)id * 2 + 2 > table_.size()) grow();
        return id;
    }
    int32_t find(const char *p, size_t n) const
    {
        const uint64_t h = hash(p, n);
        size_t mask = table_.size() - 1, i = (size_t)h & mask;
        for (;;) {
            const int32_t id = table_[i];
            if (id < 0) return -1;
            if (hashes_[(size_t)id] == h && len_[(size_t)id] == n && memcmp(arena_.data() + off_[(size_t)id], p, n) == 0) return id;
            i = (i + 1) & mask;
        }
    }
    int32_t size() const { return (int32_t)off_.size(); }
    const char *key_ptr(int32_t id) const { return arena_.data() + off_[(size_t)id]; }
    size_t key_len(int32_t id) const { return len_[(size_t)id]; }
    std::string key(int32_t id) const { return std::string(arena_.data() + off_[(size_t)id], len_[(size_t)id]); }

  private:
    std::vector<char> arena_;
    std::vector<size_t> off_;
    std::vector<uint32_t> len_;
    std::vector<uint64_t> hashes_;
    std::vector<int32_t> table_;
    static uint64_t hash(const char *p, size_t n)
    {
        uint64_t h = 1469598103934665603ull;                    // FNV-1a, finalised
        for (size_t i = 0; i < n; i++) { h ^= (unsigned char)p[i]; h *= 1099511628211ull; }
        h ^= h >> 29; h *= 0xbf58476d1ce4e5b9ull; h ^= h >> 32;
        return h;
    }
    void grow()
    {
        std::vector<int32_t> t(table_.size() * 4, -1);
        const size_t mask = t.size() - 1;
        for (int32_t id = 0; id < (int32_t)off_.size(); id++) {
            size_t i = (size_t)hashes_[(size_t)id] & mask;
            while (t[i] >= 0) i = (i + 1) & mask;
            t[i] = id;
        }
        table_.swap(t);
    }
};

// int32 -> int32 (global key id -> partition-local id), open addressing
class IntMap {
  public:
    IntMap() { keys_.assign(256, -1); vals_.assign(256, 0); }
    int32_t *find(int32_t k)
    {
        size_t mask = keys_.size() - 1, i = mix(k) & mask;
        while (keys_[i] >= 0) { if (keys_[i] == k) return &vals_[i]; i = (i + 1) & mask; }
        return nullptr;
    }
    void insert(int32_t k, int32_t v)
    {
        if ((n_ + 1) * 2 > keys_.size()) rehash();
        size_t mask = keys_.size() - 1, i = mix(k) & mask;
        while (keys_[i] >= 0) i = (i + 1) & mask;
        keys_[i] = k; vals_[i] = v; n_++;
    }

  private:
    std::vector<int32_t> keys_, vals_;
    size_t n_ = 0;
    static size_t mix(int32_t k) { uint64_t x = (uint32_t)k; x *= 0x9e3779b97f4a7c15ull; return (size_t)(x >> 20); }
    void rehash()
    {
        std::vector<int32_t> k2(keys_.size() * 4, -1), v2(keys_.size() * 4, 0);
        const size_t mask = k2.size() - 1;
        for (size_t j = 0; j < keys_.size(); j++) if (keys_[j] >= 0) {
            size_t i = mix(keys_[j]) & mask;
            while (k2[i] >= 0) i = (i + 1) & mask;
            k2[i] = keys_[j]; v2[i] = vals_[j];
        }
        keys_.swap(k2); vals_.swap(v2);
    }
};

struct PrepareOptions {                 // jobs/RegressionPrepare.java:43-46,64-69
    int num_blocks = 0;
    std::string map_key;                // "" -> random key (Math.random() in the reference; seeded here)
    bool binary_feature = false;
    int num_click_replicates = 1;
    uint64_t seed = 0;
    bool short_feature_index = false;
};

struct PartitionData {
    int pid = 0;
    std::vector<int64_t> row_ptr{0};
    std::vector<int32_t> col;           // local ids, sorted per row at finish(); intercept not stored
    std::vector<float> val;             // empty for binary.feature
    std::vector<int8_t> y;
    std::vector<float> weight, offset;
    std::vector<int32_t> l2g;           // filled by finish(): local -> global, intercept last
    IntMap g2l;                         // global key id -> local id (first-seen order)
    std::vector<int32_t> local_global;  // local id -> global id (without intercept)
    int32_t n_local() const { return (int32_t)local_global.size() + 1; }
    int32_t rows() const { return (int32_t)y.size(); }
};

struct Dataset {
    int num_blocks = 0;
    bool binary = false;
    std::vector<PartitionData> parts;
    std::vector<std::string> names;                       // global id -> name key (name [+ U+0001 + term])
    std::unordered_map<std::string, int32_t> gindex;
    int32_t n_global() const { return (int32_t)names.size() + 1; }
    int64_t total_rows() const;
};

// One decoded input row, raw (any Pig-style schema) or prepared (RegressionPrepareOutput).
struct InputRow {
    bool has_click = false, has_response = false, has_label = false, response_is_int = false;
    int click = 0, response = 0, label = 0;
    bool has_weight = false, has_offset = false, weight_is_float = false, weight_is_integral = false;
    double weight = 1.0, offset = 0.0;
    bool has_key = false;
    std::string key;                                      // map.key field or prepared "key"
    std::vector<std::pair<std::string, double>> feats;    // (name key, value as getDoubleAvro yields); value NaN = null
    std::vector<std::pair<int32_t, double>> feat_ids;      // same, keys interned (filled INSTEAD of feats when an interner is given)
    const uint8_t *raw = nullptr;                         // the record's avro encoding (valid inside the callback only)
    size_t raw_len = 0;
};

// Streams the rows of an avro file / directory; `key_field` = the field to read into InputRow::key ("" = none).
// With an interner the feature keys are interned while decoding (InputRow::feat_ids) and no strings are built per row.
void read_input_rows(const std::string &path, const std::string &key_field, bool need_values,
                     const std::function<void(InputRow &)> &fn, KeyInterner *interner = nullptr);

class DatasetBuilder {
  public:
    explicit DatasetBuilder(const PrepareOptions &opt);
    // RAW rows: RegressionPrepare semantics, then indexing.
    void add_raw(InputRow &row);
    // PREPARED rows (tmp-data): key/response/features/weight/offset already normalised.
    void add_prepared(const InputRow &row);
    Dataset finish();
    KeyInterner *interner() { return &keys_; }          // pass to read_input_rows for the allocation-free path

  private:
    PrepareOptions opt_;
    Dataset ds_;
    std::mt19937_64 rng_;
    KeyInterner keys_;                                  // global ids in first-seen order == Dataset::names at finish()
    int32_t icpt_key_ = -2;                             // interned id of "(INTERCEPT)" once seen
    // rows are queued per partition and indexed in parallel over partitions (a partition's local first-seen order
    // depends on its own rows only); flush() runs every FLUSH_ROWS rows and at finish()
    struct PendingRow { int response; float weight, offset; size_t f0, f1; };
    std::vector<std::vector<PendingRow>> pend_rows_;
    std::vector<std::vector<std::pair<int32_t, double>>> pend_feats_;
    size_t pending_ = 0;
    void intern_strings(InputRow &row);
    void add_to_partition(int pid, int response, const std::vector<std::pair<int32_t, double>> &feats, float weight, float offset);
    void index_row(PartitionData &p, const PendingRow &r, const std::pair<int32_t, double> *feats, std::vector<size_t> &ord,
                   std::vector<int32_t> &c2, std::vector<float> &v2);
    void flush();
};

struct TestRowsData {                   // jobs/RegressionAdmmTrain.java:766-811 inputs, GLOBAL feature ids
    std::vector<int64_t> row_ptr{0};
    std::vector<int32_t> gidx;
    std::vector<double> val;             // as Util.getDoubleAvro yields: evalInstanceAvro does not cast to float (models/LinearModel.java:530-534)
    std::vector<int8_t> response;
    std::vector<double> weight, offset;
    double n = 0;                       // sum of Double.parseDouble(weight.toString())
};
TestRowsData build_test_rows(const std::string &first_file, const Dataset &ds, bool binary_feature, int64_t max_rows = 1000000);

int resolve_response(const InputRow &r);   // utils/Util.java:309-337

}  // namespace mlh
