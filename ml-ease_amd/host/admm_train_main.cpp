// admm_train_main.cpp -- `mlease_admm_train <job file>`: the native host of the drop-in.
//
// Mirrors the flow of jobs/Regression.java:37-80 for the ADMM path on local files:
//   Prepare (jobs/RegressionPrepare.java:96-191, in memory; `write.tmp.data=true` also writes <out>/tmp-data)
//   -> AdmmTrain (jobs/RegressionAdmmTrain.java:130-522) with the per-iteration body behind the C-ABI
//      (include/mlease_admm.h): rows are indexed and uploaded ONCE, every iteration is one mlx_admm_iterate.
// Same .job keys and defaults, same output files: <out>/lambda-rho, <out>/sample-test-loglik/iteration-i.avro,
// <out>/best-model/best-iteration-i.avro, <out>/final-model/part-r-00000.avro.
// Extra keys (not in the reference): `gpus` (device list, default "0"), `prepared.input=true` (input.paths already
// holds RegressionPrepareOutput rows = what AdmmTrain alone consumes), `random.seed`, `write.tmp.data`.
#include <sys/stat.h>

#include <chrono>
#include <filesystem>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mlease_admm.h"
#include "avro_io.h"
#include "dataset_builder.h"
#include "java_compat.h"

using namespace mlh;

namespace {

struct Fail : std::runtime_error { using std::runtime_error::runtime_error; };

void ck(mlx_handle h, int rc, const char *what)
{
    if (rc != MLX_OK) throw Fail(std::string("Model fitting error! ") + what + ": " + mlx_last_error(h));   // jobs/...:713-716
}

void write_model_record(AvroFileWriter &w, const std::string &key, const float *z, const Dataset &ds)
{
    // LinearModel.toAvro (models/LinearModel.java:697-720): intercept first, then name / term split on U+0001
    w.put_string(key);
    w.array_start(ds.n_global());
    w.put_string("(INTERCEPT)"); w.put_string(""); w.put_float(z[ds.n_global() - 1]);
    for (int32_t j = 0; j + 1 < ds.n_global(); j++) {
        const std::string &nm = ds.names[(size_t)j];
        size_t sep = nm.find('\x01');
        w.put_string(sep == std::string::npos ? nm : nm.substr(0, sep));
        w.put_string(sep == std::string::npos ? "" : nm.substr(sep + 1));
        w.put_float(z[j]);
    }
    w.array_end();
    w.end_record();
}

void write_tmp_data(const std::string &path, const Dataset &ds)
{
    AvroFileWriter w(path, kPrepareOutputSchemaJson);
    for (auto &p : ds.parts)
        for (int32_t i = 0; i < p.rows(); i++) {
            w.put_string(std::to_string(p.pid));
            w.put_long(p.y[(size_t)i] == 1 ? 1 : 0);
            const int64_t k0 = p.row_ptr[(size_t)i], k1 = p.row_ptr[(size_t)i + 1];
            w.array_start(k1 - k0);
            for (int64_t k = k0; k < k1; k++) {
                const std::string &nm = ds.names[(size_t)p.local_global[(size_t)p.col[(size_t)k]]];
                size_t sep = nm.find('\x01');
                w.put_string(sep == std::string::npos ? nm : nm.substr(0, sep));
                w.put_string(sep == std::string::npos ? "" : nm.substr(sep + 1));
                w.put_float(ds.binary ? 1.0f : p.val[(size_t)k]);
            }
            w.array_end();
            w.put_float(p.weight[(size_t)i]);
            w.put_float(p.offset[(size_t)i]);
            w.end_record();
        }
    w.close();
}

std::vector<float> read_lambda_map(const std::string &path, const Dataset &ds)
{
    // consumers/ReadLambdaMapConsumer.java:33-53 -> dense per-feature lambda, NaN = not listed
    std::vector<float> lm((size_t)ds.n_global(), std::nanf(""));
    for (auto &file : list_avro_files(path)) {
        AvroFileReader rd(file);
        const AvroSchema &top = rd.schema();
        rd.for_each([&](AvroCursor &c) {
            std::string name, term;
            bool hn = false, hv = false;
            double v = 0;
            for (auto &f : top.fields) {
                const AvroSchema *r = c.resolve(*f.second);
                if (!r) continue;
                if (f.first == "name" && r->type == AvroType::String) { c.read_string(name); hn = true; }
                else if (f.first == "term" && r->type == AvroType::String) c.read_string(term);
                else if (f.first == "value") { v = c.read_number(*r); hv = true; }
                else c.skip(*r);
            }
            if (hn && hv) {
                if (!term.empty()) name += '\x01' + term;
                auto g = ds.gindex.find(name);
                if (g != ds.gindex.end()) lm[(size_t)g->second] = (float)v;
            }
        });
    }
    return lm;
}

bool path_exists(const std::string &p) { struct stat st; return !p.empty() && stat(p.c_str(), &st) == 0; }

}  // namespace

int run_job(const JobConfig &props)
{
    using clk = std::chrono::steady_clock;
    auto t_start = clk::now();
    const std::string out = props.get_string("output.base.path");
    const int nblocks = props.get_int("num.blocks");
    const int niter = props.get_int("num.iters", 10);                                       // :139
    const bool aggressive = props.get_bool("aggressive.liblinear.epsilon.decay", false);
    const int reg = props.get_int("regularizer");
    if (reg != 1 && reg != 2) throw Fail("Only L1 and L2 regularization supported!");       // :143-147
    const int nrep = props.get_int("num.click.replicates", 1);
    const bool binary = props.get_bool("binary.feature", false);
    const float boost = props.get_float("initialize.boost.rate", 0);
    const float rho_adapt = props.get_float("rho.adapt.coefficient", 0);
    // lambda -> rho (:153-185); a HashMap<Float,Float>: duplicates collapse; sorted ascending for the solver (:636-638)
    std::vector<std::string> lstr = props.get_list("lambda", ','), rstr = props.get_list("rho", ',');
    if (lstr.empty()) throw Fail("Undefined property: lambda");
    if (!rstr.empty() && rstr.size() != lstr.size())
        throw Fail("The number of rho's should be exactly the same as the number of lambda's. OR: don't claim rho!");
    std::map<float, float> lr;
    for (size_t j = 0; j < lstr.size(); j++) {
        const float l = strtof(lstr[j].c_str(), nullptr);
        lr[l] = rstr.empty() ? (l <= 100 ? 1.0f : 10.0f) : strtof(rstr[j].c_str(), nullptr);
    }
    std::vector<float> lam, rho;
    for (auto &kv : lr) { lam.push_back(kv.first); rho.push_back(kv.second); }
    const int nl = (int)lam.size();

    // ---- rows: Prepare + indexing, once
    PrepareOptions po;
    po.num_blocks = nblocks;
    po.map_key = props.get_string("map.key", "");
    po.binary_feature = binary;
    po.num_click_replicates = nrep;
    po.seed = (uint64_t)props.get_long("random.seed", 0);
    po.short_feature_index = props.get_bool("short.feature.index", false);
    const bool prepared = props.get_bool("prepared.input", false);
    DatasetBuilder builder(po);
    const std::string input = props.get_string("input.paths");
    if (prepared) read_input_rows(input, "key", true, [&](InputRow &r) { builder.add_prepared(r); }, builder.interner());
    else read_input_rows(input, po.map_key, !binary, [&](InputRow &r) { builder.add_raw(r); }, builder.interner());
    Dataset ds = builder.finish();
    const int ng = ds.n_global();
    auto t_indexed = clk::now();
    fprintf(stderr, "[mlease] %lld rows, %d features, %d partitions indexed in %.2f s\n", (long long)ds.total_rows(), ng - 1,
            nblocks, std::chrono::duration<double>(t_indexed - t_start).count());
    if (props.get_bool("write.tmp.data", false)) write_tmp_data(out + "/tmp-data/part-00000.avro", ds);

    std::vector<float> lambda_map;
    const std::string lm_path = props.get_string("lambda.map", "");
    if (!lm_path.empty()) lambda_map = read_lambda_map(lm_path, ds);

    {   // lambda-rho file (:200-201,721-734)
        AvroFileWriter w(out + "/lambda-rho/part-r-00000.avro",
                         "{\"type\":\"record\",\"doc\":\"The map of lambda values to rho values\",\"name\":\"LambdaRhoMap\","
                         "\"namespace\":\"com.linkedin.mlease.regression.avro\",\"fields\":[{\"name\":\"lambda\",\"type\":\"float\"},"
                         "{\"name\":\"rho\",\"type\":\"float\"}]}");
        for (int i = 0; i < nl; i++) { w.put_float(lam[(size_t)i]); w.put_float(rho[(size_t)i]); w.end_record(); }
        w.close();
    }

    // ---- devices: partition k -> device k mod G
    std::vector<int> devs;
    for (auto &s : props.get_list("gpus", ',')) devs.push_back(atoi(s.c_str()));
    if (devs.empty()) devs.push_back(0);
    const int G = (int)devs.size();
    std::vector<mlx_handle> hs((size_t)G, nullptr);
    std::vector<std::vector<int32_t>> pids_of((size_t)G);          // handle g, local index (add order) -> partition id
    char uid[MLX_UNIQUE_ID_BYTES];
    if (G > 1 && mlx_comm_get_unique_id(uid) != MLX_OK) throw Fail(std::string("RCCL: ") + mlx_last_error(nullptr));
    for (int g = 0; g < G; g++) {
        if (mlx_create(devs[(size_t)g], &hs[(size_t)g]) != MLX_OK) throw Fail(std::string("mlx_create: ") + mlx_last_error(nullptr));
        // mlease.numerics = reference_order (the DEFAULT of this drop-in since round 6) | fast. reference_order: every reduction a sequential
        // loop as in the Java code (include/mlease_admm.h: mlx_set_numerics) -- a job that replaces the reference's AdmmTrain gets the
        // reference's coefficients (north_star: within 1e-5; here bit for bit up to exp / log1p) unless it asks for the faster contract
        // (parallel trees: 1.6-2.3x the throughput, inside the reference's own row-order envelope but not within 1e-5 of it at the end of
        // the epsilon schedule). The drop-in's own key, no counterpart in the reference's job files; the LIBRARY's default
        // stays fast (mlx_create). mlease.option.<key> = <value> passes any other mlx_set_option key through.
        const std::string numerics = props.get_string("mlease.numerics", "reference_order");
        ck(hs[(size_t)g], mlx_set_option(hs[(size_t)g], "numerics", numerics.c_str()), "mlx_set_option(numerics)");
        for (const char *key : {"tick_streams", "grid_rounded_dots", "one_launch_small", "trace"}) {
            const std::string v = props.get_string(std::string("mlease.option.") + key, "");
            if (!v.empty()) ck(hs[(size_t)g], mlx_set_option(hs[(size_t)g], key, v.c_str()), "mlx_set_option");
        }
        mlx_handle h = hs[(size_t)g];
        ck(h, mlx_set_problem(h, ng, nl, lam.data(), rho.data(), nblocks, props.get_bool("penalize.intercept", false) ? 1 : 0,
                              lambda_map.empty() ? nullptr : lambda_map.data()), "mlx_set_problem");
        ck(h, mlx_set_regularizer(h, reg), "mlx_set_regularizer");
        std::vector<int32_t> a_pid, a_l, a_nl;
        std::vector<int64_t> a_nnz;
        std::vector<const int64_t *> a_rp;
        std::vector<const int32_t *> a_ci, a_l2g;
        std::vector<const float *> a_val, a_w, a_o;
        std::vector<const int8_t *> a_y;
        for (auto &p : ds.parts) {
            if (p.pid % G != g) continue;
            if (p.rows() == 0) throw Fail("Some models failed! partition " + std::to_string(p.pid) + " received no rows");   // utils/LinearModelUtils.java:80-83
            a_pid.push_back(p.pid); a_l.push_back(p.rows()); a_nl.push_back(p.n_local()); a_nnz.push_back((int64_t)p.col.size());
            a_rp.push_back(p.row_ptr.data()); a_ci.push_back(p.col.data()); a_val.push_back(ds.binary ? nullptr : p.val.data());
            a_y.push_back(p.y.data()); a_w.push_back(p.weight.data()); a_o.push_back(p.offset.data()); a_l2g.push_back(p.l2g.data());
        }
        ck(h, mlx_add_partitions_csr(h, (int32_t)a_pid.size(), a_pid.data(), a_l.data(), a_nl.data(), a_nnz.data(), a_rp.data(), a_ci.data(),
                                     ds.binary ? nullptr : a_val.data(), a_y.data(), a_w.data(), a_o.data(), a_l2g.data()),
           "mlx_add_partitions_csr");
        ck(h, mlx_finalize(h), "mlx_finalize");
        pids_of[(size_t)g] = a_pid;
    }
    if (G > 1) {
        std::vector<std::thread> th;
        std::vector<int> rcs((size_t)G, 0);
        for (int g = 0; g < G; g++) th.emplace_back([&, g] { rcs[(size_t)g] = mlx_comm_init(hs[(size_t)g], uid, G, g); });
        for (auto &t : th) t.join();
        for (int g = 0; g < G; g++) ck(hs[(size_t)g], rcs[(size_t)g], "mlx_comm_init");
    }

    // ---- test rows (:203-232): first file under test.path
    bool test_loglik = false;
    double test_n = 0;
    const std::string test_path = props.get_string("test.path", "");
    if (path_exists(test_path)) {
        auto files = list_avro_files(test_path);
        if (!files.empty()) {
            TestRowsData t = build_test_rows(files[0], ds, binary);
            if (!t.response.empty()) {
                ck(hs[0], mlx_set_test_data(hs[0], (int32_t)t.response.size(), (int64_t)t.gidx.size(), t.row_ptr.data(), t.gidx.data(),
                                            binary ? nullptr : t.val.data(), t.response.data(), t.weight.data(), t.offset.data()),
                   "mlx_set_test_data");
                test_loglik = true;
                test_n = t.n;
            }
        }
    }
    auto t_uploaded = clk::now();

    // a previous run's per-iteration outputs must not survive into this one (the reference starts from a fresh output.base.path)
    {
        std::error_code ec;
        std::filesystem::remove_all(out + "/best-model", ec);
        std::filesystem::remove_all(out + "/sample-test-loglik", ec);
        for (int it = 1; it <= niter; it++) std::filesystem::remove_all(out + "/iter-" + std::to_string(it), ec);
    }
    // ---- the loop (:278-497)
    double mindiff = 99999999;
    float liblinear_eps = 0.01f;                                                            // :279
    float best_loglik = -9999999;                                                           // :234
    const double epsilon = props.get_double("epsilon", 0.0001);
    std::vector<double> zd((size_t)nl * ng), lls((size_t)nl);
    std::vector<float> zf((size_t)nl * ng);
    int i;
    double solve_s = 0;
    auto update_loglik_best_model = [&](int it) {                                            // :812-845
        ck(hs[0], mlx_test_loglik(hs[0], lls.data()), "mlx_test_loglik");
        AvroFileWriter w(out + "/sample-test-loglik/iteration-" + std::to_string(it) + ".avro", kSampleTestLoglikSchemaJson);
        for (int li = 0; li < nl; li++) {
            const double ll = lls[(size_t)li] / test_n;
            const std::string key = java_float_to_string(lam[(size_t)li]);
            w.put_string(key); w.put_long(it); w.put_float((float)ll);
            w.end_record();
            fprintf(stderr, "[mlease] Sample test loglik for lambda=%s is: %.10g\n", key.c_str(), ll);
            if (ll > (double)best_loglik && it > 0) {
                ck(hs[0], mlx_get_z(hs[0], nullptr, zf.data()), "mlx_get_z");
                // fs.delete(outBasePath + "/best-model", true) before every write (:838-840): exactly ONE model file exists
                std::error_code ec;
                std::filesystem::remove_all(out + "/best-model", ec);
                AvroFileWriter bw(out + "/best-model/best-iteration-" + std::to_string(it) + ".avro", kLinearModelSchemaJson);
                write_model_record(bw, key, zf.data() + (size_t)li * ng, ds);
                bw.close();
                best_loglik = (float)ll;
            }
        }
        w.close();
    };
    auto on_all_devices = [&](const char *what, const std::function<int(int)> &call) {
        if (G == 1) { ck(hs[0], call(0), what); return; }
        std::vector<std::thread> th;
        std::vector<int> rcs((size_t)G, 0);
        for (int g = 0; g < G; g++) th.emplace_back([&, g] { rcs[(size_t)g] = call(g); });
        for (auto &t : th) t.join();
        for (int g = 0; g < G; g++) ck(hs[(size_t)g], rcs[(size_t)g], what);
    };
    // write.iter.files=true: the per-iteration files the reference leaves under output.base.path/iter-<i>/ (jobs/...:309-334; kept
    // when remove.tmp.dir=false): u (the u_k every reducer of iteration i reads; an EMPTY model file at i = 1, :310-312),
    // init-value (z as the solvers see it, float32) and, after the iteration, model: one RegressionTrainOutput record per reduce key
    // "<lambda>#<partition>" with the reducer's beta_k and u_k + beta_k (:641-718, avro/RegressionTrainOutput.avsc:17-39).
    // Vectors are written dense over the global feature list (the reference writes the keys of its HashMaps: absent ones read as 0).
    const bool write_iter = props.get_bool("write.iter.files", false);
    bool warm_start_done = false;                                   // z is the empty model before iteration 1 unless the mean-model warm start ran
    std::vector<float> pm_b((size_t)ng), pm_x((size_t)ng), pm_u((size_t)ng);
    auto reduce_key = [&](int li, int32_t pid) { return java_float_to_string(lam[(size_t)li]) + "#" + std::to_string(pid); };
    auto write_iter_inputs = [&](int it) {
        const std::string dir = out + "/iter-" + std::to_string(it);
        {
            AvroFileWriter w(dir + "/u/part-r-00000.avro", kLinearModelSchemaJson);
            if (it > 1)
                for (int g = 0; g < G; g++)
                    for (size_t k = 0; k < pids_of[(size_t)g].size(); k++)
                        for (int li = 0; li < nl; li++) {
                            ck(hs[(size_t)g], mlx_get_partition_model(hs[(size_t)g], (int32_t)k, li, nullptr, nullptr, pm_u.data()), "mlx_get_partition_model");
                            write_model_record(w, reduce_key(li, pids_of[(size_t)g][k]), pm_u.data(), ds);
                        }
            w.close();
        }
        ck(hs[0], mlx_get_z(hs[0], nullptr, zf.data()), "mlx_get_z");
        AvroFileWriter w(dir + "/init-value/part-r-00000.avro", kLinearModelSchemaJson);
        if (it > 1 || warm_start_done)
            for (int li = 0; li < nl; li++) write_model_record(w, java_float_to_string(lam[(size_t)li]), zf.data() + (size_t)li * ng, ds);
        w.close();
    };
    auto write_iter_models = [&](int it) {
        AvroFileWriter w(out + "/iter-" + std::to_string(it) + "/model/part-r-00000.avro", kTrainOutputSchemaJson);
        for (int g = 0; g < G; g++)
            for (size_t k = 0; k < pids_of[(size_t)g].size(); k++)
                for (int li = 0; li < nl; li++) {
                    ck(hs[(size_t)g], mlx_get_partition_model(hs[(size_t)g], (int32_t)k, li, pm_b.data(), pm_x.data(), nullptr), "mlx_get_partition_model");
                    w.put_string(reduce_key(li, pids_of[(size_t)g][k]));
                    for (const float *v : {pm_b.data(), pm_x.data()}) {
                        w.array_start(ng);
                        w.put_string("(INTERCEPT)"); w.put_string(""); w.put_float(v[ng - 1]);
                        for (int32_t j = 0; j + 1 < ng; j++) {
                            const std::string &nm = ds.names[(size_t)j];
                            size_t sep = nm.find('\x01');
                            w.put_string(sep == std::string::npos ? nm : nm.substr(0, sep));
                            w.put_string(sep == std::string::npos ? "" : nm.substr(sep + 1));
                            w.put_float(v[j]);
                        }
                        w.array_end();
                    }
                    w.end_record();
                }
        w.close();
    };
    const bool warm_start = boost > 0 && reg == 2;
    if (warm_start) {
        // Initialize z by the mean model (:236-276): RegressionNaiveTrain on the same partitions with the job's
        // liblinear.epsilon (0.01 when unset, :246-249) and prior.mean, then z = meanModel(...)
        const double ieps = float_string_roundtrip(props.get_float("liblinear.epsilon", 0.01f));
        const double pmean = (double)props.get_float("prior.mean", 0.0f);
        auto t0 = clk::now();
        on_all_devices("mlx_naive_init", [&](int g) { return mlx_naive_init(hs[(size_t)g], ieps, pmean, nullptr); });
        solve_s += std::chrono::duration<double>(clk::now() - t0).count();
        fprintf(stderr, "[mlease] mean model initialised (liblinear epsilon %g)\n", ieps);
        if (test_loglik) update_loglik_best_model(0);                                        // :271-274
        warm_start_done = true;
    }
    for (i = 1; i <= niter; i++) {
        float rate = 1.0f;
        if (i == 1 && warm_start) rate = boost;                                              // :313-317
        if (i > 1 && rho_adapt > 0) rate = (float)std::exp((double)(-(i - 1) * rho_adapt)); // :323-327 (float product, double exp)
        if (i > 1 && mindiff < 0.001 && !aggressive) liblinear_eps = liblinear_eps / 10;     // :338-341
        else if (aggressive && i > 5) liblinear_eps = liblinear_eps / 10;                    // :342-345
        const double eps = float_string_roundtrip(liblinear_eps);                            // :346,:620,:702
        if (write_iter) write_iter_inputs(i);
        auto t0 = clk::now();
        std::vector<mlx_stats> st((size_t)G);
        on_all_devices("mlx_admm_iterate", [&](int g) { return mlx_admm_iterate(hs[(size_t)g], eps, rate, &st[(size_t)g]); });
        solve_s += std::chrono::duration<double>(clk::now() - t0).count();
        const double maxdiff = st[0].maxdiff;
        mindiff = st[0].mindiff;
        fprintf(stderr, "[mlease] iteration %d: liblinear epsilon %s, max |z - z_prev| = %.6g, min = %.6g\n", i,
                java_float_to_string(liblinear_eps).c_str(), maxdiff, mindiff);
        if (write_iter) write_iter_models(i);
        if (test_loglik) update_loglik_best_model(i);
        if (maxdiff < epsilon && liblinear_eps <= 0.00001) break;                            // :493-496
    }
    // final-model (:499-501)
    ck(hs[0], mlx_get_z(hs[0], zd.data(), zf.data()), "mlx_get_z");
    {
        AvroFileWriter w(out + "/final-model/part-r-00000.avro", kLinearModelSchemaJson);
        for (int li = 0; li < nl; li++) write_model_record(w, java_float_to_string(lam[(size_t)li]), zf.data() + (size_t)li * ng, ds);
        w.close();
    }
    {
        // Which numerics contract produced this model (nothing in the outputs said): the run log, and a side file next to
        // lambda-rho whose leading underscore hides it from Hadoop's input listings (FileInputFormat skips "_*" and ".*"), so a consumer of
        // final-model / best-model sees the reference's files only.
        char nbuf[64] = "", kbuf[64] = "", tbuf[32] = "";
        mlx_get_option(hs[0], "numerics", nbuf, sizeof nbuf);
        mlx_get_option(hs[0], "numerics_kernels", kbuf, sizeof kbuf);
        mlx_get_option(hs[0], "dense_tiles", tbuf, sizeof tbuf);
        fprintf(stderr, "[mlease] numerics contract: %s (kernels: %s; dense tiles on device 0: %s) -- %s\n", nbuf, kbuf, tbuf,
                std::string(nbuf) == "fast" ? "parallel reduction trees; job key mlease.numerics=reference_order (the default) runs the reference's sequential sums bit for bit"
                                            : "every reduction in the reference's order; job key mlease.numerics=fast selects the parallel-tree contract (1.6-2.3x the throughput)");
        FILE *mf = fopen((out + "/_mlease_run.json").c_str(), "w");
        if (mf) {
            fprintf(mf, "{\"library\": \"%s\", \"numerics\": \"%s\", \"numerics_kernels\": \"%s\", \"gpus\": %d, \"admm_iterations\": %d, \"num_blocks\": %d, \"lambdas\": %d}\n",
                    mlx_version(), nbuf, kbuf, G, std::min(i, niter), nblocks, nl);
            fclose(mf);
        }
    }
    for (auto h : hs) mlx_destroy(h);
    fprintf(stderr, "[mlease] done: index %.2f s, upload %.2f s, %d ADMM iterations in %.3f s, total %.2f s\n",
            std::chrono::duration<double>(t_indexed - t_start).count(), std::chrono::duration<double>(t_uploaded - t_indexed).count(),
            std::min(i, niter), solve_s, std::chrono::duration<double>(clk::now() - t_start).count());
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 2) {
        fprintf(stderr, "[Usage]: mlease_admm_train <Job config path> [key=value ...]\n");    // jobs/Regression.java:90-94
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
