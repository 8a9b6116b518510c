// avro_io.h -- native Avro object-container reader/writer for the formats around the ADMM hot path
// (SURVEY.md 8f N1). Replaces, for local files, mapred/AvroFileReader.java:56-82 + the generated
// RegressionPrepareOutput / GenericData.Record decoding the reference does per row per iteration.
//
// Reader: any writer schema (records, nullable unions as Pig writes them, arrays, maps, enums, fixed);
// values are pulled through a small cursor API so that no per-row tree is materialised.
// Codecs: null, deflate (zlib). Writer: record schemas given as JSON text, deflate level 9 like
// mapred/AbstractAvroJob.java:253.
#pragma once
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <cstdio>
#include <string>
#include <vector>

#include "json_min.h"

namespace mlh {

enum class AvroType { Null, Boolean, Int, Long, Float, Double, Bytes, String, Record, Enum, Array, Map, Union, Fixed };

struct AvroSchema {
    AvroType type = AvroType::Null;
    std::string name;                                    // record / enum / fixed
    std::vector<std::pair<std::string, std::shared_ptr<AvroSchema>>> fields;   // record
    std::shared_ptr<AvroSchema> items;                   // array items / map values
    std::vector<std::shared_ptr<AvroSchema>> branches;   // union
    std::vector<std::string> symbols;                    // enum
    int fixed_size = 0;
    int field_index(const std::string &n) const;
};

std::shared_ptr<AvroSchema> parse_schema(const std::string &json_text);

// Cursor over one decoded block.
class AvroCursor {
  public:
    AvroCursor(const uint8_t *p, const uint8_t *end) : p_(p), end_(end) {}
    int64_t read_long();
    float read_float();
    double read_double();
    bool read_bool();
    void read_string(std::string &out);
    void skip_bytes();
    void skip(const AvroSchema &s);
    // Resolve a (possibly nullable-union) node to its concrete branch; returns nullptr when the value is null.
    const AvroSchema *resolve(const AvroSchema &s);
    // Numeric value of a resolved int/long/float/double/boolean node as the Java Number.doubleValue() would give.
    double read_number(const AvroSchema &resolved);
    bool at_end() const { return p_ >= end_; }
    const uint8_t *ptr() const { return p_; }

  private:
    const uint8_t *p_, *end_;
    void need(size_t n) const;
};

class AvroFileReader {
  public:
    explicit AvroFileReader(const std::string &path);
    ~AvroFileReader();
    AvroFileReader(const AvroFileReader &) = delete;
    AvroFileReader &operator=(const AvroFileReader &) = delete;
    const AvroSchema &schema() const { return *schema_; }
    const std::string &schema_json() const { return schema_json_; }
    // Calls fn(cursor) once per record; fn must consume exactly one record of schema().
    void for_each(const std::function<void(AvroCursor &)> &fn);
    // Block-level access for parallel decoding: the (possibly deflated) payload spans in file order, and the inflater.
    struct RawBlock { int64_t count; const uint8_t *begin, *end; };
    std::vector<RawBlock> blocks() const;
    void inflate(const RawBlock &b, std::vector<uint8_t> &out) const;      // out = the block's datum bytes
    bool deflated() const { return codec_ == "deflate"; }

  private:
    const uint8_t *base_ = nullptr;      // the mapped file
    size_t size_ = 0;
    size_t pos_ = 0;
    std::string codec_, schema_json_;
    uint8_t sync_[16];
    std::shared_ptr<AvroSchema> schema_;
};

// part files of a directory (name order) or the file itself (utils/Util.java findPartFiles / AvroUtils.enumerateFiles)
std::vector<std::string> list_avro_files(const std::string &path);

class AvroFileWriter {
  public:
    AvroFileWriter(const std::string &path, const std::string &schema_json, const std::string &codec = "deflate");
    ~AvroFileWriter();
    // encoders append to the current record buffer
    void put_long(int64_t v);
    void put_float(float v);
    void put_double(double v);
    void put_string(const std::string &s);
    void put_raw(const uint8_t *p, size_t n);   // already-encoded bytes (e.g. the fields of an input record, verbatim)
    void array_start(int64_t count);      // count items follow, then array_end()
    void array_end();
    void end_record();                    // one datum complete
    void close();

  private:
    std::string path_, codec_;
    std::vector<uint8_t> block_, file_;
    int64_t block_count_ = 0;
    bool closed_ = false;
    FILE *out_ = nullptr;
    void flush_block();
    void write_out();
};

extern const char *kLinearModelSchemaJson;        // avro/LinearModelAvro.avsc:16-31
extern const char *kPrepareOutputSchemaJson;      // avro/RegressionPrepareOutput.avsc:16-34
extern const char *kSampleTestLoglikSchemaJson;   // avro/SampleTestLoglik.avsc:16-26
extern const char *kTrainOutputSchemaJson;        // avro/RegressionTrainOutput.avsc:17-39

}  // namespace mlh
