// regression_test_main.cpp -- `mlease_regression_test <job file> [key=value ...]`: the scoring job of the drop-in.
//
// Mirrors com.linkedin.mlease.regression.jobs.RegressionTest (jobs/RegressionTest.java:64-232) for local files:
//   for every lambda of the job: pred = (float) model_lambda.evalInstanceAvro(record, false, binary.feature) for every
//   record of input.paths (AdmmTestMapper :147-175), written to output.base.path/lambda-<lambda>/part-r-00000.avro as the
//   input fields + `pred` (schema AdmmTestOutput, :201-232), in ascending pred order (the shuffle sorts the Float key,
//   AdmmTestReducer :178-198 just re-emits); then once more with model.base.path/best-model when that exists (:94-107).
// The dot products run on the GPU (mlx_score_rows); the input records are copied through byte for byte.
#include <sys/stat.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/mlease_admm.h"
#include "avro_io.h"
#include "dataset_builder.h"
#include "java_compat.h"

using namespace mlh;

namespace {

struct Fail : std::runtime_error { using std::runtime_error::runtime_error; };

struct Model {
    std::string key;
    std::vector<std::string> names;     // feature keys (name U+0001 term), file order
    std::vector<float> coef;            // names.size() + 1, intercept last
};

// consumers/ReadLinearModelConsumer + models/LinearModel.java:112-156: every record of every part file; a later record
// with the same key replaces an earlier one (HashMap.put)
std::vector<Model> read_model_file(const std::string &path)
{
    std::vector<Model> out;
    for (auto &file : list_avro_files(path)) {
        AvroFileReader rd(file);
        const AvroSchema &top = rd.schema();
        rd.for_each([&](AvroCursor &c) {
            Model m;
            float icpt = 0;
            for (auto &f : top.fields) {
                const AvroSchema *r = c.resolve(*f.second);
                if (!r) continue;
                if (f.first == "key" && r->type == AvroType::String) c.read_string(m.key);
                else if (f.first == "model" && r->type == AvroType::Array) {
                    const AvroSchema &item = *r->items;
                    for (;;) {
                        int64_t cnt = c.read_long();
                        if (cnt == 0) break;
                        if (cnt < 0) { cnt = -cnt; c.read_long(); }
                        for (int64_t i = 0; i < cnt; i++) {
                            std::string name, term;
                            double v = 0;
                            for (auto &g : item.fields) {
                                const AvroSchema *q = c.resolve(*g.second);
                                if (!q) continue;
                                if (g.first == "name" && q->type == AvroType::String) c.read_string(name);
                                else if (g.first == "term" && q->type == AvroType::String) c.read_string(term);
                                else if (g.first == "value") v = c.read_number(*q);
                                else c.skip(*q);
                            }
                            if (!term.empty()) name += '\x01' + term;
                            if (name == "(INTERCEPT)") icpt = (float)v;
                            else { m.names.push_back(name); m.coef.push_back((float)v); }
                        }
                    }
                } else c.skip(*r);
            }
            m.coef.push_back(icpt);
            bool replaced = false;
            for (auto &o : out) if (o.key == m.key) { o = m; replaced = true; }
            if (!replaced) out.push_back(std::move(m));
        });
    }
    return out;
}

bool is_dir(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode); }

// jobs/RegressionTest.java:201-232: the input record's fields + pred, named AdmmTestOutput
std::string output_schema_json(const std::string &input_schema_json)
{
    Json in = JsonParser(input_schema_json).parse();
    const Json *rec = &in;
    if (in.is_arr()) {                                     // Util.removeUnion
        rec = nullptr;
        for (auto &b : in.arr) if (b.is_obj() && b.get("type") && b.get("type")->is_str() && b.get("type")->str == "record") rec = &b;
        if (!rec) throw Fail("Input does not have schema info and/or input is missing.");
    }
    const Json *fields = rec->get("fields");
    if (!fields || !fields->is_arr()) throw Fail("input schema is not a record");
    Json out;
    out.kind = Json::Obj;
    auto str = [](const std::string &s) { Json j; j.kind = Json::Str; j.str = s; return j; };
    out.obj.emplace_back("type", str("record"));
    out.obj.emplace_back("name", str("AdmmTestOutput"));
    out.obj.emplace_back("namespace", str("com.linkedin.lab.regression.avro"));
    out.obj.emplace_back("doc", str("Test output for AdmmTest"));
    Json fl;
    fl.kind = Json::Arr;
    for (auto &f : fields->arr) {
        Json g;
        g.kind = Json::Obj;
        for (auto &kv : f.obj) if (kv.first == "name" || kv.first == "type" || kv.first == "doc") g.obj.push_back(kv);
        fl.arr.push_back(g);
    }
    Json pred;
    pred.kind = Json::Obj;
    pred.obj.emplace_back("name", str("pred"));
    pred.obj.emplace_back("type", str("float"));
    pred.obj.emplace_back("doc", str(""));
    fl.arr.push_back(pred);
    out.obj.emplace_back("fields", fl);
    std::string text;
    json_dump(out, text);
    return text;
}

}  // namespace

int run_job(const JobConfig &props)
{
    const std::string input = props.get_string("input.paths", "");
    if (input.empty()) {
        fprintf(stderr, "[mlease] test.input.paths is empty! So no test will be done!\n");       // :109-111
        return 0;
    }
    const std::string out = props.get_string("output.base.path"), model_base = props.get_string("model.base.path");
    const bool binary = props.get_bool("binary.feature", false);
    std::vector<std::string> lambdas = props.get_list("lambda", ',');

    // rows: raw bytes kept for the pass-through, features as (name key, value) resolved per model below
    struct Row { size_t raw_off, raw_len, f0, f1; double offset; };
    std::vector<Row> rows;
    std::vector<uint8_t> raw;
    std::vector<std::pair<std::string, double>> feats;
    read_input_rows(input, "", !binary, [&](InputRow &r) {
        const int y = resolve_response(r);                                                       // evalInstanceAvro :497-499
        if (y != 1 && y != 0 && y != -1) throw Fail("response = " + std::to_string(y));
        Row w{raw.size(), r.raw_len, feats.size(), 0, r.has_offset ? r.offset : 0.0};
        raw.insert(raw.end(), r.raw, r.raw + r.raw_len);
        for (auto &f : r.feats) {
            if (!binary && std::isnan(f.second)) throw Fail("value is null");
            feats.emplace_back(f.first, binary ? 1.0 : f.second);
        }
        w.f1 = feats.size();
        rows.push_back(w);
    });
    const std::string schema_json = output_schema_json(AvroFileReader(list_avro_files(input).at(0)).schema_json());
    const int32_t l = (int32_t)rows.size();
    fprintf(stderr, "[mlease] %d test rows, %zu feature entries\n", l, feats.size());

    mlx_handle h = nullptr;
    if (mlx_create(props.get_int("gpu", 0), &h) != MLX_OK) throw Fail(std::string("mlx_create: ") + mlx_last_error(nullptr));
    auto score_and_write = [&](const Model &m, const std::string &dir) {
        std::unordered_map<std::string, int32_t> index;
        for (size_t j = 0; j < m.names.size(); j++) index[m.names[j]] = (int32_t)j;              // later duplicates win like HashMap.put
        std::vector<int64_t> rp((size_t)l + 1, 0);
        std::vector<int32_t> gi(feats.size());
        std::vector<double> val(binary ? 0 : feats.size());     // doubles as evalInstanceAvro reads them (models/LinearModel.java:530-534)
        std::vector<double> off((size_t)l);
        for (int32_t i = 0; i < l; i++) {
            for (size_t k = rows[(size_t)i].f0; k < rows[(size_t)i].f1; k++) {
                auto it = index.find(feats[k].first);
                gi[k] = it == index.end() ? -1 : it->second;
                if (!binary) val[k] = feats[k].second;
            }
            rp[(size_t)i + 1] = (int64_t)rows[(size_t)i].f1;
            off[(size_t)i] = rows[(size_t)i].offset;
        }
        std::vector<float> pred((size_t)l);
        if (mlx_score_rows(h, (int32_t)m.coef.size(), m.coef.data(), l, (int64_t)feats.size(), rp.data(), gi.data(),
                           binary ? nullptr : val.data(), off.data(), pred.data()) != MLX_OK)
            throw Fail(std::string("mlx_score_rows: ") + mlx_last_error(h));
        std::vector<int32_t> order((size_t)l);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return pred[(size_t)a] < pred[(size_t)b]; });
        AvroFileWriter w(dir + "/part-r-00000.avro", schema_json);
        for (int32_t i : order) {
            w.put_raw(raw.data() + rows[(size_t)i].raw_off, rows[(size_t)i].raw_len);
            w.put_float(pred[(size_t)i]);
            w.end_record();
        }
        w.close();
        fprintf(stderr, "[mlease] wrote %s/part-r-00000.avro (model %s, %zu coefficients)\n", dir.c_str(), m.key.c_str(), m.coef.size());
    };
    const std::vector<Model> models = read_model_file(model_base + "/final-model");
    fprintf(stderr, "[mlease] Loaded the model for test, size:%zu\n", models.size());           // :139
    for (const std::string &lam : lambdas) {
        const std::string key = java_float_to_string(strtof(lam.c_str(), nullptr));              // String.valueOf(_lambda), :157
        const Model *m = nullptr;
        for (auto &x : models) if (x.key == key) m = &x;
        if (!m) throw Fail("no model for lambda " + key + " in " + model_base + "/final-model");
        score_and_write(*m, out + "/lambda-" + lam);
    }
    if (is_dir(model_base + "/best-model") && !list_avro_files(model_base + "/best-model").empty()) {
        // the training job keeps exactly one best-iteration-<i>.avro (it deletes the directory before each write,
        // jobs/RegressionAdmmTrain.java:838-840); several files mean a foreign or stale directory: refuse rather than score a wrong model
        const std::vector<std::string> files = list_avro_files(model_base + "/best-model");
        if (files.size() > 1) throw Fail("more than one model file under " + model_base + "/best-model (" + std::to_string(files.size()) + "): expected a single best-iteration-<i>.avro");
        const std::vector<Model> best = read_model_file(model_base + "/best-model");
        if (!best.empty()) score_and_write(best.front(), out + "/best-model");
    }
    mlx_destroy(h);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 2) {
        fprintf(stderr, "[Usage]: mlease_regression_test <Job config path> [key=value ...]\n");
        return 2;
    }
    try {
        JobConfig cfg = JobConfig::from_file(argv[1]);
        for (int a = 2; a < argc; a++) {
            const char *eq = strchr(argv[a], '=');
            if (eq) cfg.put(std::string(argv[a], (size_t)(eq - argv[a])), eq + 1);
        }
        return run_job(cfg);
    } catch (const std::exception &e) {
        fprintf(stderr, "[mlease] ERROR: %s\n", e.what());
        return 1;
    }
}
