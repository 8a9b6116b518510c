// dataset_builder.h -- native row preparation + partition indexing.
//
// Mirrors  RegressionPrepare.RegressionPrepareMapper.map      jobs/RegressionPrepare.java:96-191
//          AdmmPartitioner.getPartition                       jobs/RegressionAdmmTrain.java:579-590
//          LibLinearDataset.addInstanceAvro + finish          liblinearfunc/LibLinearDataset.java:413-484,586-658
//          LibLinearBinaryDataset.addInstanceAvro             liblinearfunc/LibLinearBinaryDataset.java:426-515
// and produces, per partition, exactly the arrays mlx_add_partition_csr takes (include/mlease_admm.h).
#pragma once
#include <cstdint>
#include <random>
#include <string>
#include <unordered_map>
#include <vector>

#include "avro_io.h"

namespace mlh {

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
    std::unordered_map<std::string, int32_t> index;   // name key -> local id (first-seen order)
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
    const uint8_t *raw = nullptr;                         // the record's avro encoding (valid inside the callback only)
    size_t raw_len = 0;
};

// Streams the rows of an avro file / directory; `key_field` = the field to read into InputRow::key ("" = none).
void read_input_rows(const std::string &path, const std::string &key_field, bool need_values,
                     const std::function<void(InputRow &)> &fn);

class DatasetBuilder {
  public:
    explicit DatasetBuilder(const PrepareOptions &opt);
    // RAW rows: RegressionPrepare semantics, then indexing.
    void add_raw(InputRow &row);
    // PREPARED rows (tmp-data): key/response/features/weight/offset already normalised.
    void add_prepared(const InputRow &row);
    Dataset finish();

  private:
    PrepareOptions opt_;
    Dataset ds_;
    std::mt19937_64 rng_;
    void add_to_partition(int pid, int response, const std::vector<std::pair<std::string, double>> &feats, float weight, float offset);
};

struct TestRowsData {                   // jobs/RegressionAdmmTrain.java:766-811 inputs, GLOBAL feature ids
    std::vector<int64_t> row_ptr{0};
    std::vector<int32_t> gidx;
    std::vector<float> val;
    std::vector<int8_t> response;
    std::vector<double> weight, offset;
    double n = 0;                       // sum of Double.parseDouble(weight.toString())
};
TestRowsData build_test_rows(const std::string &first_file, const Dataset &ds, bool binary_feature, int64_t max_rows = 1000000);

int resolve_response(const InputRow &r);   // utils/Util.java:309-337

}  // namespace mlh
