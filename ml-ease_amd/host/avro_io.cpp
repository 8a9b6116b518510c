// avro_io.cpp -- see avro_io.h
#include "avro_io.h"

#include <dirent.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cstring>
#include <fstream>
#include <stdexcept>

namespace mlh {

const char *kLinearModelSchemaJson =
    "{\"type\":\"record\",\"doc\":\"Linear Model in Avro format\",\"name\":\"LinearModelAvro\",\"namespace\":\"com.linkedin.mlease.avro\","
    "\"fields\":[{\"name\":\"key\",\"type\":\"string\"},{\"name\":\"model\",\"type\":{\"type\":\"array\",\"items\":{\"type\":\"record\","
    "\"name\":\"feature\",\"fields\":[{\"name\":\"name\",\"type\":\"string\"},{\"name\":\"term\",\"type\":\"string\"},"
    "{\"name\":\"value\",\"type\":\"float\"}]}}}]}";
const char *kPrepareOutputSchemaJson =
    "{\"type\":\"record\",\"doc\":\"Output for RegressionPrepare job before running Regression train jobs\","
    "\"name\":\"RegressionPrepareOutput\",\"namespace\":\"com.linkedin.mlease.regression.avro\",\"fields\":["
    "{\"name\":\"key\",\"type\":\"string\"},{\"name\":\"response\",\"type\":\"int\"},{\"name\":\"features\",\"type\":{\"type\":\"array\","
    "\"items\":{\"type\":\"record\",\"name\":\"feature\",\"fields\":[{\"name\":\"name\",\"type\":\"string\"},{\"name\":\"term\",\"type\":\"string\"},"
    "{\"name\":\"value\",\"type\":\"float\"}]}}},{\"name\":\"weight\",\"type\":\"float\"},{\"name\":\"offset\",\"type\":\"float\"}]}";
const char *kSampleTestLoglikSchemaJson =
    "{\"type\":\"record\",\"doc\":\"Sample test loglik over iterations\",\"name\":\"SampleTestLoglik\","
    "\"namespace\":\"com.linkedin.mlease.regression.avro\",\"fields\":[{\"name\":\"lambda\",\"type\":\"string\"},"
    "{\"name\":\"iter\",\"type\":\"int\"},{\"name\":\"testLoglik\",\"type\":\"float\"}]}";
const char *kTrainOutputSchemaJson =
    "{\"type\":\"record\",\"doc\":\"Model output from AdmmTrain\",\"name\":\"RegressionTrainOutput\","
    "\"namespace\":\"com.linkedin.mlease.regression.avro\",\"fields\":[{\"name\":\"key\",\"type\":\"string\"},"
    "{\"name\":\"model\",\"type\":{\"type\":\"array\",\"items\":{\"type\":\"record\",\"name\":\"feature\",\"fields\":["
    "{\"name\":\"name\",\"type\":\"string\"},{\"name\":\"term\",\"type\":\"string\"},{\"name\":\"value\",\"type\":\"float\"}]}}},"
    "{\"name\":\"uplusx\",\"type\":{\"type\":\"array\",\"items\":{\"type\":\"record\",\"name\":\"feature1\",\"fields\":["
    "{\"name\":\"name\",\"type\":\"string\"},{\"name\":\"term\",\"type\":\"string\"},{\"name\":\"value\",\"type\":\"float\"}]}}}]}";

int AvroSchema::field_index(const std::string &n) const
{
    for (size_t i = 0; i < fields.size(); i++)
        if (fields[i].first == n) return (int)i;
    return -1;
}

namespace {

using Named = std::map<std::string, std::shared_ptr<AvroSchema>>;

std::shared_ptr<AvroSchema> build(const Json &j, Named &named)
{
    auto s = std::make_shared<AvroSchema>();
    if (j.is_str()) {
        const std::string &t = j.str;
        if (t == "null") s->type = AvroType::Null;
        else if (t == "boolean") s->type = AvroType::Boolean;
        else if (t == "int") s->type = AvroType::Int;
        else if (t == "long") s->type = AvroType::Long;
        else if (t == "float") s->type = AvroType::Float;
        else if (t == "double") s->type = AvroType::Double;
        else if (t == "bytes") s->type = AvroType::Bytes;
        else if (t == "string") s->type = AvroType::String;
        else {
            auto it = named.find(t);
            if (it == named.end()) {
                auto dot = t.rfind('.');
                if (dot != std::string::npos) it = named.find(t.substr(dot + 1));
            }
            if (it == named.end()) throw std::runtime_error("avro: unknown type " + t);
            return it->second;
        }
        return s;
    }
    if (j.is_arr()) {
        s->type = AvroType::Union;
        for (auto &b : j.arr) s->branches.push_back(build(b, named));
        return s;
    }
    if (!j.is_obj()) throw std::runtime_error("avro: bad schema node");
    const Json *t = j.get("type");
    if (!t) throw std::runtime_error("avro: schema object without type");
    if (!t->is_str()) return build(*t, named);
    const std::string &ts = t->str;
    if (ts == "record") {
        s->type = AvroType::Record;
        if (const Json *n = j.get("name")) s->name = n->str;
        named[s->name] = s;
        if (const Json *ns = j.get("namespace")) named[ns->str + "." + s->name] = s;
        const Json *f = j.get("fields");
        if (!f || !f->is_arr()) throw std::runtime_error("avro: record without fields");
        for (auto &fd : f->arr) {
            const Json *fn = fd.get("name"), *ft = fd.get("type");
            if (!fn || !ft) throw std::runtime_error("avro: bad field");
            s->fields.emplace_back(fn->str, build(*ft, named));
        }
    } else if (ts == "array") {
        s->type = AvroType::Array;
        s->items = build(*j.get("items"), named);
    } else if (ts == "map") {
        s->type = AvroType::Map;
        s->items = build(*j.get("values"), named);
    } else if (ts == "enum") {
        s->type = AvroType::Enum;
        if (const Json *n = j.get("name")) { s->name = n->str; named[s->name] = s; }
        for (auto &sy : j.get("symbols")->arr) s->symbols.push_back(sy.str);
    } else if (ts == "fixed") {
        s->type = AvroType::Fixed;
        if (const Json *n = j.get("name")) { s->name = n->str; named[s->name] = s; }
        s->fixed_size = (int)j.get("size")->num;
    } else {
        return build(*t, named);      // {"type": "string"} and friends
    }
    return s;
}

}  // namespace

std::shared_ptr<AvroSchema> parse_schema(const std::string &json_text)
{
    Json j = JsonParser(json_text).parse();
    Named named;
    return build(j, named);
}

// --------------------------------------------------------------------------------- cursor
void AvroCursor::need(size_t n) const
{
    if ((size_t)(end_ - p_) < n) throw std::runtime_error("avro: truncated data");
}
int64_t AvroCursor::read_long()
{
    uint64_t acc = 0;
    int shift = 0;
    for (;;) {
        need(1);
        uint8_t b = *p_++;
        acc |= (uint64_t)(b & 0x7F) << shift;
        if (!(b & 0x80)) break;
        shift += 7;
        if (shift > 63) throw std::runtime_error("avro: varint too long");
    }
    return (int64_t)(acc >> 1) ^ -(int64_t)(acc & 1);
}
float AvroCursor::read_float() { need(4); float v; memcpy(&v, p_, 4); p_ += 4; return v; }
double AvroCursor::read_double() { need(8); double v; memcpy(&v, p_, 8); p_ += 8; return v; }
bool AvroCursor::read_bool() { need(1); return *p_++ != 0; }
void AvroCursor::read_string(std::string &out)
{
    int64_t n = read_long();
    if (n < 0) throw std::runtime_error("avro: negative string length");
    need((size_t)n);
    out.assign((const char *)p_, (size_t)n);
    p_ += n;
}
void AvroCursor::skip_bytes()
{
    int64_t n = read_long();
    need((size_t)n);
    p_ += n;
}
void AvroCursor::skip(const AvroSchema &s)
{
    switch (s.type) {
    case AvroType::Null: break;
    case AvroType::Boolean: need(1); p_++; break;
    case AvroType::Int: case AvroType::Long: case AvroType::Enum: read_long(); break;
    case AvroType::Float: need(4); p_ += 4; break;
    case AvroType::Double: need(8); p_ += 8; break;
    case AvroType::Bytes: case AvroType::String: skip_bytes(); break;
    case AvroType::Fixed: need((size_t)s.fixed_size); p_ += s.fixed_size; break;
    case AvroType::Record: for (auto &f : s.fields) skip(*f.second); break;
    case AvroType::Union: { int64_t b = read_long(); skip(*s.branches.at((size_t)b)); break; }
    case AvroType::Array: case AvroType::Map:
        for (;;) {
            int64_t n = read_long();
            if (n == 0) break;
            if (n < 0) { int64_t bytes = read_long(); need((size_t)bytes); p_ += bytes; continue; }
            for (int64_t i = 0; i < n; i++) {
                if (s.type == AvroType::Map) skip_bytes();
                skip(*s.items);
            }
        }
        break;
    }
}
const AvroSchema *AvroCursor::resolve(const AvroSchema &s)
{
    const AvroSchema *cur = &s;
    while (cur->type == AvroType::Union) {
        int64_t b = read_long();
        cur = cur->branches.at((size_t)b).get();
    }
    return cur->type == AvroType::Null ? nullptr : cur;
}
double AvroCursor::read_number(const AvroSchema &r)
{
    switch (r.type) {
    case AvroType::Int: case AvroType::Long: return (double)read_long();
    case AvroType::Float: return (double)read_float();
    case AvroType::Double: return read_double();
    case AvroType::Boolean: return read_bool() ? 1.0 : 0.0;
    default: throw std::runtime_error("avro: value is not a number");
    }
}

// --------------------------------------------------------------------------------- reader
AvroFileReader::AvroFileReader(const std::string &path)
{
    // the file is mapped, not copied: blocks are inflated / decoded straight out of the page cache
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("cannot open " + path);
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); throw std::runtime_error("cannot stat " + path); }
    size_ = (size_t)st.st_size;
    if (size_ > 0) {
        void *m = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { close(fd); throw std::runtime_error("cannot map " + path); }
        base_ = static_cast<const uint8_t *>(m);
    }
    close(fd);
    struct Unmap {                          // the constructor can still throw: release the mapping then (the destructor does not run)
        AvroFileReader *r; bool armed = true;
        ~Unmap() { if (armed && r->base_ && r->size_) { munmap(const_cast<uint8_t *>(r->base_), r->size_); r->base_ = nullptr; r->size_ = 0; } }
    } guard{this};
    if (size_ < 4 || memcmp(base_, "Obj\x01", 4) != 0) throw std::runtime_error(path + " is not an avro object container file");
    AvroCursor c(base_ + 4, base_ + size_);
    codec_ = "null";
    for (;;) {                                      // metadata map<bytes>
        int64_t n = c.read_long();
        if (n == 0) break;
        if (n < 0) { n = -n; c.read_long(); }
        for (int64_t i = 0; i < n; i++) {
            std::string k, v;
            c.read_string(k);
            c.read_string(v);
            if (k == "avro.schema") schema_json_ = v;
            else if (k == "avro.codec") codec_ = v;
        }
    }
    if (codec_ != "null" && codec_ != "deflate") throw std::runtime_error("unsupported avro codec " + codec_);
    const size_t consumed = (size_t)(c.ptr() - base_);
    if (consumed + 16 > size_) throw std::runtime_error("avro: truncated header");
    memcpy(sync_, base_ + consumed, 16);
    pos_ = consumed + 16;
    schema_ = parse_schema(schema_json_);
    guard.armed = false;
}

AvroFileReader::~AvroFileReader() { if (base_ && size_) munmap(const_cast<uint8_t *>(base_), size_); }

std::vector<AvroFileReader::RawBlock> AvroFileReader::blocks() const
{
    std::vector<RawBlock> out;
    size_t pos = pos_;
    while (pos < size_) {
        AvroCursor hc(base_ + pos, base_ + size_);
        const int64_t count = hc.read_long(), size = hc.read_long();
        const uint8_t *q = hc.ptr();
        if (size < 0 || q + size + 16 > base_ + size_) throw std::runtime_error("avro: bad block size");
        if (memcmp(q + size, sync_, 16) != 0) throw std::runtime_error("avro: sync marker mismatch");
        out.push_back({count, q, q + size});
        pos = (size_t)(q + size + 16 - base_);
    }
    return out;
}

void AvroFileReader::inflate(const RawBlock &b, std::vector<uint8_t> &out) const
{
    out.clear();
    if (codec_ != "deflate") { out.assign(b.begin, b.end); return; }
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) throw std::runtime_error("zlib init failed");
    zs.next_in = const_cast<Bytef *>(b.begin);
    zs.avail_in = (uInt)(b.end - b.begin);
    out.resize(std::max<size_t>((size_t)(b.end - b.begin) * 4, 1 << 16));
    size_t have = 0;
    int rc;
    do {
        if (have == out.size()) out.resize(out.size() * 2);
        zs.next_out = out.data() + have;
        zs.avail_out = (uInt)std::min<size_t>(out.size() - have, 1u << 30);
        const size_t before = zs.avail_out;
        rc = ::inflate(&zs, Z_NO_FLUSH);
        if (rc != Z_OK && rc != Z_STREAM_END) { inflateEnd(&zs); throw std::runtime_error("avro: inflate failed"); }
        have += before - zs.avail_out;
    } while (rc != Z_STREAM_END);
    inflateEnd(&zs);
    out.resize(have);
}

void AvroFileReader::for_each(const std::function<void(AvroCursor &)> &fn)
{
    std::vector<uint8_t> inflated;
    for (const RawBlock &b : blocks()) {
        const uint8_t *bp = b.begin, *be = b.end;
        if (deflated()) { inflate(b, inflated); bp = inflated.data(); be = inflated.data() + inflated.size(); }
        AvroCursor c(bp, be);
        for (int64_t i = 0; i < b.count; i++) fn(c);
    }
}

std::vector<std::string> list_avro_files(const std::string &path)
{
    struct stat st;
    if (stat(path.c_str(), &st) != 0) throw std::runtime_error("no such path: " + path);
    std::vector<std::string> out;
    if (S_ISDIR(st.st_mode)) {
        DIR *d = opendir(path.c_str());
        if (!d) throw std::runtime_error("cannot list " + path);
        while (dirent *e = readdir(d)) {
            std::string n = e->d_name;
            if (n.size() > 5 && n.substr(n.size() - 5) == ".avro") out.push_back(path + "/" + n);
        }
        closedir(d);
        std::sort(out.begin(), out.end());
    } else out.push_back(path);
    return out;
}

// --------------------------------------------------------------------------------- writer
static void put_varint(std::vector<uint8_t> &b, int64_t v)
{
    uint64_t n = ((uint64_t)v << 1) ^ (uint64_t)(v >> 63);
    while (n & ~0x7FULL) { b.push_back((uint8_t)((n & 0x7F) | 0x80)); n >>= 7; }
    b.push_back((uint8_t)n);
}
static const uint8_t kSync[16] = {'m', 'l', 'e', 'a', 's', 'e', '-', 'a', 'm', 'd', '-', 's', 'y', 'n', 'c', '!'};

AvroFileWriter::AvroFileWriter(const std::string &path, const std::string &schema_json, const std::string &codec)
    : path_(path), codec_(codec)
{
    file_.insert(file_.end(), {'O', 'b', 'j', 1});
    put_varint(file_, 2);
    auto put_str = [&](const std::string &s) { put_varint(file_, (int64_t)s.size()); file_.insert(file_.end(), s.begin(), s.end()); };
    put_str("avro.schema"); put_str(schema_json);
    put_str("avro.codec"); put_str(codec_);
    put_varint(file_, 0);
    file_.insert(file_.end(), kSync, kSync + 16);
}
AvroFileWriter::~AvroFileWriter()
{
    // callers close() explicitly and see its exceptions; a writer dropped on an error path must not throw from here
    if (!closed_) { try { close(); } catch (...) {} }
    if (out_) fclose(out_);
}
void AvroFileWriter::put_long(int64_t v) { put_varint(block_, v); }
void AvroFileWriter::put_float(float v) { uint8_t b[4]; memcpy(b, &v, 4); block_.insert(block_.end(), b, b + 4); }
void AvroFileWriter::put_double(double v) { uint8_t b[8]; memcpy(b, &v, 8); block_.insert(block_.end(), b, b + 8); }
void AvroFileWriter::put_string(const std::string &s) { put_varint(block_, (int64_t)s.size()); block_.insert(block_.end(), s.begin(), s.end()); }
void AvroFileWriter::put_raw(const uint8_t *p, size_t n) { block_.insert(block_.end(), p, p + n); }
void AvroFileWriter::array_start(int64_t count) { if (count > 0) put_varint(block_, count); }
void AvroFileWriter::array_end() { put_varint(block_, 0); }
void AvroFileWriter::end_record() { if (++block_count_ >= 4096 || block_.size() > (8u << 20)) flush_block(); }
void AvroFileWriter::flush_block()
{
    if (block_count_ == 0) return;
    std::vector<uint8_t> payload;
    if (codec_ == "deflate") {
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, 9, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw std::runtime_error("zlib init failed");
        payload.resize(deflateBound(&zs, (uLong)block_.size()));
        zs.next_in = block_.data(); zs.avail_in = (uInt)block_.size();
        zs.next_out = payload.data(); zs.avail_out = (uInt)payload.size();
        if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { deflateEnd(&zs); throw std::runtime_error("deflate failed"); }
        payload.resize(zs.total_out);
        deflateEnd(&zs);
    } else payload.swap(block_);
    put_varint(file_, block_count_);
    put_varint(file_, (int64_t)payload.size());
    file_.insert(file_.end(), payload.begin(), payload.end());
    file_.insert(file_.end(), kSync, kSync + 16);
    block_.clear();
    block_count_ = 0;
    if (file_.size() > (8u << 20)) write_out();
}
static void mkdirs_for(const std::string &path)
{
    for (size_t i = 1; i < path.size(); i++)
        if (path[i] == '/') mkdir(path.substr(0, i).c_str(), 0777);
}
// Finished blocks are streamed to the file (at most ~8 MiB stay in memory) and every write is checked: a full disk or
// an IO error raises instead of leaving a truncated container behind an exit code 0.
void AvroFileWriter::write_out()
{
    if (file_.empty()) return;
    if (!out_) {
        mkdirs_for(path_);
        out_ = fopen(path_.c_str(), "wb");
        if (!out_) throw std::runtime_error("cannot write " + path_);
    }
    if (fwrite(file_.data(), 1, file_.size(), out_) != file_.size()) throw std::runtime_error("write failed: " + path_);
    file_.clear();
}
void AvroFileWriter::close()
{
    if (closed_) return;
    flush_block();
    write_out();
    if (!out_) {                         // nothing buffered can only mean an empty header -- still create the file
        mkdirs_for(path_);
        out_ = fopen(path_.c_str(), "wb");
        if (!out_) throw std::runtime_error("cannot write " + path_);
    }
    closed_ = true;
    FILE *f = out_;
    out_ = nullptr;
    if (fflush(f) != 0 || ferror(f)) { fclose(f); throw std::runtime_error("write failed: " + path_); }
    if (fclose(f) != 0) throw std::runtime_error("close failed: " + path_);
}

}  // namespace mlh
