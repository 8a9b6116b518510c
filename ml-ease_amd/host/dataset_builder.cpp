// dataset_builder.cpp -- see dataset_builder.h
#include "dataset_builder.h"

#include <atomic>
#include <thread>
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <stdexcept>

#include "java_compat.h"

namespace mlh {

static const char *kInterceptName = "(INTERCEPT)";      // liblinearfunc/LibLinearDataset.java:92

int64_t Dataset::total_rows() const
{
    int64_t n = 0;
    for (auto &p : parts) n += p.rows();
    return n;
}

// --------------------------------------------------------------------------------- row decoding
namespace {

struct RowPlan {
    const AvroSchema *rec = nullptr;
    int f_features = -1, f_response = -1, f_click = -1, f_label = -1, f_weight = -1, f_offset = -1, f_key = -1;
};

void read_flag(AvroCursor &c, const AvroSchema &s, bool &has, int &out, bool *is_int)
{
    const AvroSchema *r = c.resolve(s);
    if (!r) { has = false; return; }
    has = true;
    if (r->type == AvroType::Boolean) { out = c.read_bool() ? 1 : 0; if (is_int) *is_int = false; }
    else if (r->type == AvroType::Int) { out = (int)c.read_long(); if (is_int) *is_int = true; }
    else throw std::runtime_error("Response/Click/Label column should be either boolean or int32!");   // utils/Util.java:322-325
}

double read_double_like(AvroCursor &c, const AvroSchema &r, bool *is_float = nullptr, bool *is_integral = nullptr)
{
    if (is_float) *is_float = (r.type == AvroType::Float);
    if (is_integral) *is_integral = (r.type == AvroType::Int || r.type == AvroType::Long);
    if (r.type == AvroType::String) {                     // Util.getDoubleAvro: String -> atof (utils/Util.java:85-88)
        std::string s;
        c.read_string(s);
        if (s.empty()) throw std::runtime_error("Can't convert empty string to double");
        return strtod(s.c_str(), nullptr);
    }
    return c.read_number(r);
}

// Which member of the feature record is what (resolved once per file instead of three string compares per non-zero)
struct ItemPlan {
    const AvroSchema *items = nullptr;
    std::vector<int> kind;                 // per field of the item record: 0 skip, 1 name, 2 term, 3 value
};

void read_features(AvroCursor &c, const AvroSchema &s, bool need_values, ItemPlan &plan, KeyInterner *interner,
                   std::string &name, std::string &term, InputRow &row)
{
    row.feats.clear();
    row.feat_ids.clear();
    const AvroSchema *arr = c.resolve(s);
    if (!arr) throw std::runtime_error("features is null");
    if (arr->type != AvroType::Array) throw std::runtime_error("features is not a list");
    int idx = 0;
    for (;;) {
        int64_t n = c.read_long();
        if (n == 0) break;
        if (n < 0) { n = -n; c.read_long(); }
        for (int64_t i = 0; i < n; i++, idx++) {
            const AvroSchema *it = c.resolve(*arr->items);
            if (!it || it->type != AvroType::Record) throw std::runtime_error("features[" + std::to_string(idx) + "] is not a record");
            if (plan.items != it) {
                plan.items = it;
                plan.kind.clear();
                for (auto &f : it->fields) plan.kind.push_back(f.first == "name" ? 1 : f.first == "term" ? 2 : f.first == "value" ? 3 : 0);
            }
            bool have_name = false;
            double value = std::nan("");
            term.clear();
            for (size_t k = 0; k < it->fields.size(); k++) {
                const AvroSchema &fs = *it->fields[k].second;
                switch (plan.kind[k]) {
                case 1: {
                    const AvroSchema *r = c.resolve(fs);
                    if (r) { if (r->type != AvroType::String) throw std::runtime_error("name is not a string"); c.read_string(name); have_name = true; }
                    break;
                }
                case 2: {
                    const AvroSchema *r = c.resolve(fs);
                    if (r) { if (r->type != AvroType::String) throw std::runtime_error("term is not a string"); c.read_string(term); }
                    break;
                }
                case 3: {
                    const AvroSchema *r = c.resolve(fs);
                    if (r) { if (need_values) value = read_double_like(c, *r); else c.skip(*r); }
                    break;
                }
                default: c.skip(fs);
                }
            }
            if (!have_name) throw std::runtime_error("name is null");
            if (!term.empty()) { name += '\x01'; name += term; }          // LibLinearDataset.java:458-459
            if (interner) row.feat_ids.emplace_back(interner->intern(name.data(), name.size()), value);
            else row.feats.emplace_back(name, value);
        }
    }
}

}  // namespace

namespace {

// one record of the input -> row (feature keys as strings, or interned when `interner` is given)
void decode_row(AvroCursor &c, const AvroSchema &top, const RowPlan &pl, bool need_values, ItemPlan &iplan,
                KeyInterner *interner, std::string &name_buf, std::string &term_buf, InputRow &row)
{
    row.has_click = row.has_response = row.has_label = row.response_is_int = false;
    row.has_weight = row.has_offset = row.weight_is_float = row.weight_is_integral = row.has_key = false;
    row.weight = 1.0; row.offset = 0.0; row.click = row.response = row.label = 0;
    row.key.clear();
    row.raw = c.ptr();
    bool saw_features = false;
    for (int i = 0; i < (int)top.fields.size(); i++) {
        const AvroSchema &fs = *top.fields[i].second;
        if (i == pl.f_features) { read_features(c, fs, need_values, iplan, interner, name_buf, term_buf, row); saw_features = true; }
        else if (i == pl.f_response) read_flag(c, fs, row.has_response, row.response, &row.response_is_int);
        else if (i == pl.f_click) read_flag(c, fs, row.has_click, row.click, nullptr);
        else if (i == pl.f_label) read_flag(c, fs, row.has_label, row.label, nullptr);
        else if (i == pl.f_weight) {
            const AvroSchema *r = c.resolve(fs);
            if (r) { row.has_weight = true; row.weight = read_double_like(c, *r, &row.weight_is_float, &row.weight_is_integral); }
        } else if (i == pl.f_offset) {
            const AvroSchema *r = c.resolve(fs);
            if (r) { row.has_offset = true; row.offset = read_double_like(c, *r); }
        } else if (i == pl.f_key) {
            const AvroSchema *r = c.resolve(fs);
            if (r) {
                row.has_key = true;
                if (r->type == AvroType::String) c.read_string(row.key);
                else if (r->type == AvroType::Int || r->type == AvroType::Long) row.key = std::to_string(c.read_long());
                else throw std::runtime_error("map.key field must be a string or an integer");
            }
        } else c.skip(fs);
    }
    if (!saw_features) throw std::runtime_error("features is null");
    row.raw_len = (size_t)(c.ptr() - row.raw);
}

int loader_threads()
{
    if (const char *e = getenv("MLH_THREADS")) return std::max(1, atoi(e));
    const unsigned hc = std::thread::hardware_concurrency();
    return (int)std::min<unsigned>(hc ? hc : 1, 32);
}

}  // namespace

void read_input_rows(const std::string &path, const std::string &key_field, bool need_values,
                     const std::function<void(InputRow &)> &fn, KeyInterner *interner)
{
    const int nthreads = interner ? loader_threads() : 1;
    for (const std::string &file : list_avro_files(path)) {
        AvroFileReader rd(file);
        const AvroSchema &top = rd.schema();
        if (top.type != AvroType::Record) throw std::runtime_error(file + ": top-level schema is not a record");
        RowPlan pl;
        pl.rec = &top;
        pl.f_features = top.field_index("features");
        pl.f_response = top.field_index("response");
        pl.f_click = top.field_index("click");
        pl.f_label = top.field_index("label");
        pl.f_weight = top.field_index("weight");
        pl.f_offset = top.field_index("offset");
        pl.f_key = key_field.empty() ? -1 : top.field_index(key_field);
        const std::vector<AvroFileReader::RawBlock> blocks = rd.blocks();
        if (nthreads <= 1 || blocks.size() < 2) {
            InputRow row;
            ItemPlan iplan;
            std::string name_buf, term_buf;
            rd.for_each([&](AvroCursor &c) {
                decode_row(c, top, pl, need_values, iplan, interner, name_buf, term_buf, row);
                fn(row);
            });
            continue;
        }
        // Parallel path: inflating and decoding a block is independent of every other block; only the assignment of ids
        // in first-seen order is not. Workers decode whole blocks against a block-local interner; the caller's thread
        // then walks the blocks in file order, maps each block's new keys to global ids in the order the stream shows
        // them (== what the sequential path does) and hands the rows on.
        struct BlockOut { std::vector<uint8_t> bytes; std::vector<InputRow> rows; KeyInterner keys; std::string error; };
        const size_t batch = (size_t)nthreads * 2;
        for (size_t b0 = 0; b0 < blocks.size(); b0 += batch) {
            const size_t nb = std::min(batch, blocks.size() - b0);
            std::vector<BlockOut> outs(nb);
            std::atomic<size_t> next{0};
            auto work = [&]() {
                for (;;) {
                    const size_t j = next.fetch_add(1);
                    if (j >= nb) return;
                    BlockOut &o = outs[j];
                    try {
                        const AvroFileReader::RawBlock &rb = blocks[b0 + j];
                        rd.inflate(rb, o.bytes);
                        AvroCursor c(o.bytes.data(), o.bytes.data() + o.bytes.size());
                        ItemPlan iplan;
                        std::string name_buf, term_buf;
                        o.rows.resize((size_t)rb.count);
                        for (int64_t i = 0; i < rb.count; i++)
                            decode_row(c, top, pl, need_values, iplan, &o.keys, name_buf, term_buf, o.rows[(size_t)i]);
                    } catch (const std::exception &e) {
                        o.error = e.what();
                        if (o.error.empty()) o.error = "decode error";
                    }
                }
            };
            std::vector<std::thread> th;
            for (int t = 0; t < std::min<int>(nthreads, (int)nb) - 1; t++) th.emplace_back(work);
            work();
            for (auto &t : th) t.join();
            std::vector<int32_t> remap;
            for (size_t j = 0; j < nb; j++) {
                BlockOut &o = outs[j];
                // rows decoded before a failing one are still delivered, then the error surfaces: same as the sequential path
                remap.assign((size_t)o.keys.size(), -1);
                for (InputRow &row : o.rows) {
                    if (row.raw == nullptr) break;                      // not reached by the decoder (error in this block)
                    for (auto &f : row.feat_ids) {
                        int32_t &g = remap[(size_t)f.first];
                        if (g < 0) g = interner->intern(o.keys.key_ptr(f.first), o.keys.key_len(f.first));
                        f.first = g;
                    }
                    fn(row);
                }
                if (!o.error.empty()) throw std::runtime_error(o.error);
            }
        }
    }
}

int resolve_response(const InputRow &r)
{
    // last non-null of click, response, label (utils/Util.java:311-316)
    if (r.has_label) return r.label;
    if (r.has_response) return r.response;
    if (r.has_click) return r.click;
    throw std::runtime_error("Data should contain one field of the three: response, click or label!");
}

// --------------------------------------------------------------------------------- builder
DatasetBuilder::DatasetBuilder(const PrepareOptions &opt) : opt_(opt), rng_(opt.seed)
{
    ds_.num_blocks = opt.num_blocks;
    ds_.binary = opt.binary_feature;
    ds_.parts.resize((size_t)std::max(0, opt.num_blocks));
    for (int k = 0; k < opt.num_blocks; k++) ds_.parts[(size_t)k].pid = k;
}

void DatasetBuilder::intern_strings(InputRow &row)
{
    // callers that decoded without the interner (tests, embedding): same ids, assigned here in the same order
    if (!row.feat_ids.empty() || row.feats.empty()) return;
    row.feat_ids.reserve(row.feats.size());
    for (auto &f : row.feats) row.feat_ids.emplace_back(keys_.intern(f.first.data(), f.first.size()), f.second);
}

void DatasetBuilder::add_to_partition(int pid, int response, const std::vector<std::pair<int32_t, double>> &feats,
                                      float weight, float offset)
{
    if (pid < 0 || pid >= ds_.num_blocks)
        throw std::runtime_error("Map key is wrong! key has to be in the range of [0,numPartitions-1].");   // :585-588
    if (response != 1 && response != 0 && response != -1)
        throw std::runtime_error("response = " + std::to_string(response) + " (only 1, 0, -1 are allowed)");   // :419-420
    if (weight < 0) throw std::runtime_error("weight = " + std::to_string(weight) + " (weight cannot < 0)");     // :428-429
    if (pend_rows_.empty()) { pend_rows_.resize((size_t)ds_.num_blocks); pend_feats_.resize((size_t)ds_.num_blocks); }
    auto &pf = pend_feats_[(size_t)pid];
    const size_t f0 = pf.size();
    pf.insert(pf.end(), feats.begin(), feats.end());
    pend_rows_[(size_t)pid].push_back({response, weight, offset, f0, pf.size()});
    if (++pending_ >= 65536) flush();
}

void DatasetBuilder::index_row(PartitionData &p, const PendingRow &r, const std::pair<int32_t, double> *feats,
                               std::vector<size_t> &ord, std::vector<int32_t> &c2, std::vector<float> &v2)
{
    p.y.push_back(r.response == 1 ? 1 : -1);                                                                    // :421-423
    p.weight.push_back(r.weight);
    p.offset.push_back(r.offset);
    const size_t start = p.col.size();
    bool sorted = true;
    int32_t prev = -1;
    for (size_t k = r.f0; k < r.f1; k++) {
        const std::pair<int32_t, double> &f = feats[k];
        if (opt_.binary_feature && f.second != 1.0)
            throw std::runtime_error("Cannot handle non-binary feature value (all feature values have to be 1; or just do not specify the value)");
        int32_t id;
        if (int32_t *hit = p.g2l.find(f.first)) id = *hit;
        else {
            if (f.first == icpt_key_) throw std::runtime_error(std::string("feature name cannot be ") + kInterceptName);   // :470-471
            id = (int32_t)p.local_global.size();
            if (opt_.short_feature_index && id + 1 >= 32767)
                throw std::runtime_error("When using short to store feature indices, you cannot have more than 32766 features!!");
            p.g2l.insert(f.first, id);
            p.local_global.push_back(f.first);            // global id == interned key id (first-seen order of the stream)
        }
        if (id < prev) sorted = false;
        prev = id;
        p.col.push_back(id);
        if (!opt_.binary_feature) p.val.push_back((float)f.second);
    }
    // per-row sort by local id, stable (LibLinearDataset.java:481-482); rows usually arrive sorted already
    const size_t m = p.col.size() - start;
    if (m > 1 && !sorted) {
        ord.resize(m);
        std::iota(ord.begin(), ord.end(), 0);
        std::stable_sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return p.col[start + a] < p.col[start + b]; });
        c2.resize(m);
        if (!opt_.binary_feature) v2.resize(m);
        for (size_t i = 0; i < m; i++) { c2[i] = p.col[start + ord[i]]; if (!opt_.binary_feature) v2[i] = p.val[start + ord[i]]; }
        std::copy(c2.begin(), c2.end(), p.col.begin() + (long)start);
        if (!opt_.binary_feature) std::copy(v2.begin(), v2.end(), p.val.begin() + (long)start);
    }
    p.row_ptr.push_back((int64_t)p.col.size());
}

void DatasetBuilder::flush()
{
    if (pending_ == 0) return;
    icpt_key_ = keys_.find(kInterceptName, strlen(kInterceptName));               // -1 unless such a key was seen
    const int np = ds_.num_blocks;
    const int nthreads = std::min(loader_threads(), np);
    std::vector<std::string> errors((size_t)np);
    std::atomic<int> next{0};
    auto work = [&]() {
        std::vector<size_t> ord;
        std::vector<int32_t> c2;
        std::vector<float> v2;
        for (;;) {
            const int k = next.fetch_add(1);
            if (k >= np) return;
            try {
                for (const PendingRow &r : pend_rows_[(size_t)k]) index_row(ds_.parts[(size_t)k], r, pend_feats_[(size_t)k].data(), ord, c2, v2);
            } catch (const std::exception &e) {
                errors[(size_t)k] = e.what();
            }
            pend_rows_[(size_t)k].clear();
            pend_feats_[(size_t)k].clear();
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads - 1; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    pending_ = 0;
    for (auto &e : errors) if (!e.empty()) throw std::runtime_error(e);
}

void DatasetBuilder::add_raw(InputRow &row)
{
    std::string mapkey;
    if (!opt_.map_key.empty()) {
        if (!row.has_key) throw std::runtime_error("map.key is wrongly specified! No such key exists in some lines of the data!");   // :103-106
        mapkey = row.key;
    } else {
        const double u = std::generate_canonical<double, 53>(rng_);
        mapkey = std::to_string((int)std::floor(u * opt_.num_blocks));                    // :112
    }
    const int response = resolve_response(row);
    intern_strings(row);
    for (auto &f : row.feat_ids) {
        if (opt_.binary_feature) f.second = 1.0;                                           // :142-146
        else if (std::isnan(f.second)) throw std::runtime_error("value is null");
        else f.second = (double)(float)f.second;
    }
    double weight = row.has_weight ? row.weight : 1.0;
    if (!row.has_response) throw std::runtime_error("response is null");                  // Util.getIntAvro :159
    if (!row.response_is_int) throw std::runtime_error("response is not an integer");
    if (row.response == 1) weight = weight / opt_.num_click_replicates;                   // :159-162
    const float wf = (float)weight;
    const float of = (float)(row.has_offset ? row.offset : 0.0);
    if (opt_.map_key.empty() && response == 1) {                                          // :172-186
        int pid = atoi(mapkey.c_str());
        for (int i = 0; i < opt_.num_click_replicates; i++) {
            if (pid >= opt_.num_blocks) pid -= opt_.num_blocks;
            add_to_partition(pid, response, row.feat_ids, wf, of);
            pid++;
        }
    } else {
        char *end = nullptr;
        long pid = strtol(mapkey.c_str(), &end, 10);
        if (end == mapkey.c_str() || *end) throw std::runtime_error("For input string: \"" + mapkey + "\"");   // Integer.parseInt, :558
        add_to_partition((int)pid, response, row.feat_ids, wf, of);
    }
}

void DatasetBuilder::add_prepared(const InputRow &row)
{
    if (!row.has_key) throw std::runtime_error("prepared row without key");
    char *end = nullptr;
    long pid = strtol(row.key.c_str(), &end, 10);
    if (end == row.key.c_str() || *end) throw std::runtime_error("For input string: \"" + row.key + "\"");
    if (!row.has_response) throw std::runtime_error("prepared row without response");
    InputRow &rw = const_cast<InputRow &>(row);
    intern_strings(rw);
    for (auto &f : rw.feat_ids) if (std::isnan(f.second)) f.second = 1.0;
    add_to_partition((int)pid, row.response, rw.feat_ids, (float)(row.has_weight ? row.weight : 1.0), (float)(row.has_offset ? row.offset : 0.0));
}

Dataset DatasetBuilder::finish()
{
    flush();
    // global dictionary: the interned keys in first-seen order
    ds_.names.clear();
    ds_.gindex.clear();
    ds_.names.reserve((size_t)keys_.size());
    for (int32_t id = 0; id < keys_.size(); id++) { ds_.names.push_back(keys_.key(id)); ds_.gindex.emplace(ds_.names.back(), id); }
    const int32_t ng = ds_.n_global();
    for (auto &p : ds_.parts) {
        p.l2g = p.local_global;
        p.l2g.push_back(ng - 1);                                                          // intercept = last local / last global
    }
    return std::move(ds_);
}

// --------------------------------------------------------------------------------- test rows
TestRowsData build_test_rows(const std::string &first_file, const Dataset &ds, bool binary_feature, int64_t max_rows)
{
    TestRowsData t;
    int64_t nrec = 0;
    try {
        read_input_rows(first_file, "", !binary_feature, [&](InputRow &r) {
            if (nrec >= max_rows) throw std::length_error("max rows");                    // MAX_NTEST_EVENTS :799
            const int y = resolve_response(r);
            if (y != 1 && y != 0 && y != -1) throw std::runtime_error("response = " + std::to_string(y));
            for (auto &f : r.feats) {
                auto g = ds.gindex.find(f.first);
                t.gidx.push_back(g == ds.gindex.end() ? -1 : g->second);
                if (!binary_feature) {
                    if (std::isnan(f.second)) throw std::runtime_error("value is null");
                    t.val.push_back(f.second);
                }
            }
            t.row_ptr.push_back((int64_t)t.gidx.size());
            t.response.push_back((int8_t)y);
            t.weight.push_back(r.has_weight ? r.weight : 1.0);
            t.offset.push_back(r.has_offset ? r.offset : 0.0);
            // n += Double.parseDouble(record.get("weight").toString()) (:792-798): a Float prints shortest-float32
            if (!r.has_weight) t.n += 1.0;
            else if (r.weight_is_float) t.n += float_string_roundtrip((float)r.weight);
            else t.n += r.weight;
            nrec++;
        });
    } catch (const std::length_error &) {
    }
    return t;
}

}  // namespace mlh
