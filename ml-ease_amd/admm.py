"""Host-side mirror of ``RegressionAdmmTrain.run`` (jobs/RegressionAdmmTrain.java:130-522) for the
part of the loop that is NOT on the GPU: job-config keys and defaults, the lambda -> rho table, the
float32 liblinear-epsilon schedule with its String round trip, rho adaptation rates, the stop rule,
and the ``final-model`` file. Everything numeric per iteration happens behind the engine
(:class:`mlease_amd.hip_engine.HipAdmmEngine` -> include/mlease_admm.h).

The engine is passed in; this module never imports the CPU oracle (tests wrap the oracle in the
same small protocol to check the host logic and the sharded exchange on CPU/gloo).
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

from . import avro_io
from .dataset import INTERCEPT_NAME, TERM_SEP


# ------------------------------------------------------------------------------- Java float <-> string
def java_float_to_string(x) -> str:
    """``Float.toString`` / ``String.valueOf(float)``: shortest digits that round-trip the float32,
    plain decimal for 1e-3 <= |x| < 1e7, otherwise ``d.dddE[-]n`` (used for the model keys
    ``String.valueOf(lambda)``, jobs/RegressionAdmmTrain.java:184,650, and for liblinear epsilon, :702)."""
    f = np.float32(x)
    if np.isnan(f):
        return "NaN"
    if np.isinf(f):
        return "Infinity" if f > 0 else "-Infinity"
    if f == 0:
        return "-0.0" if np.signbit(f) else "0.0"
    s = np.format_float_scientific(f, unique=True, trim="0", exp_digits=1)     # e.g. '9.999999e-4', '1.e-2'
    mant, exp = s.split("e")
    sign = ""
    if mant.startswith("-"):
        sign, mant = "-", mant[1:]
    ip, _, fp = mant.partition(".")
    digits = (ip + fp).rstrip("0") or "0"
    e = int(exp)
    if len(digits) == 1:
        # Java renders at least two significant digits, the closest such decimal (Float.MIN_VALUE -> "1.4E-45")
        m2, e2 = ("%.1e" % abs(float(f))).split("e")
        digits, e = m2.replace(".", "").rstrip("0") or "0", int(e2)
    a = abs(float(f))
    if 1e-3 <= a < 1e7:
        if e >= 0:
            if len(digits) <= e + 1:
                return sign + digits + "0" * (e + 1 - len(digits)) + ".0"
            return sign + digits[:e + 1] + "." + digits[e + 1:]
        return sign + "0." + "0" * (-e - 1) + digits
    return sign + digits[0] + "." + (digits[1:] or "0") + "E" + str(e)


def float_string_roundtrip(x) -> float:
    """``Double.parseDouble(String.valueOf(float))`` (conf.setFloat -> "epsilon=" + ... -> Util.atof;
    jobs/RegressionAdmmTrain.java:346,620,702 ; llf/LibLinear.java:128-131)."""
    return float(java_float_to_string(x))


# ------------------------------------------------------------------------------- job config
def parse_job_file(path: str) -> Dict[str, str]:
    """``java.util.Properties`` subset used by .job files (mapred/JobConfig.java:78-90): ``k=v`` / ``k:v``,
    ``#``/``!`` comments, surrounding whitespace stripped."""
    props: Dict[str, str] = {}
    with open(path, "r", encoding="latin-1") as fh:
        for raw in fh:
            line = raw.strip()
            if not line or line[0] in "#!":
                continue
            for i, ch in enumerate(line):
                if ch in "=:":
                    props[line[:i].strip()] = line[i + 1:].strip()
                    break
            else:
                props[line] = ""
    return props


def _get_bool(p: Dict[str, str], k: str, d: bool) -> bool:
    return d if k not in p else p[k].strip().lower() == "true"        # Boolean.parseBoolean


@dataclass
class AdmmConfig:
    """Keys and defaults of jobs/RegressionAdmmTrain.java:78-120,138-151,302,473."""
    num_blocks: int
    lambdas: List[float]
    rhos: Optional[List[float]] = None
    num_iters: int = 10
    regularizer: int = 2
    epsilon: float = 1e-4
    penalize_intercept: bool = False
    binary_feature: bool = False
    short_feature_index: bool = False
    aggressive_liblinear_epsilon_decay: bool = False
    rho_adapt_coefficient: float = 0.0
    initialize_boost_rate: float = 0.0
    num_click_replicates: int = 1
    output_base_path: str = ""
    test_path: str = ""
    extras: Dict[str, str] = field(default_factory=dict)

    @classmethod
    def from_properties(cls, p: Dict[str, str]) -> "AdmmConfig":
        for req in ("output.base.path", "num.blocks", "lambda", "regularizer"):
            if req not in p:
                raise KeyError("Undefined property: " + req)             # mapred/JobConfig getString/getInt
        reg = int(p["regularizer"])
        if reg not in (1, 2):
            raise IOError("Only L1 and L2 regularization supported!")    # :143-147
        lam = [float(np.float32(s)) for s in p["lambda"].split(",") if s.strip() != ""]
        rho = None
        if "rho" in p:
            rho = [float(np.float32(s)) for s in p["rho"].split(",") if s.strip() != ""]
            if len(rho) != len(lam):
                raise IOError("The number of rho's should be exactly the same as the number of lambda's. OR: don't claim rho!")
        return cls(num_blocks=int(p["num.blocks"]), lambdas=lam, rhos=rho, num_iters=int(p.get("num.iters", 10)),
                   regularizer=reg, epsilon=float(p.get("epsilon", 0.0001)),
                   penalize_intercept=_get_bool(p, "penalize.intercept", False),
                   binary_feature=_get_bool(p, "binary.feature", False),
                   short_feature_index=_get_bool(p, "short.feature.index", False),
                   aggressive_liblinear_epsilon_decay=_get_bool(p, "aggressive.liblinear.epsilon.decay", False),
                   rho_adapt_coefficient=float(np.float32(p.get("rho.adapt.coefficient", 0))),
                   initialize_boost_rate=float(np.float32(p.get("initialize.boost.rate", 0))),
                   num_click_replicates=int(p.get("num.click.replicates", 1)),
                   output_base_path=p["output.base.path"], test_path=p.get("test.path", ""), extras=dict(p))

    def lambda_rho(self) -> Dict[np.float32, np.float32]:
        """lambda -> rho (:163-185): given, else 1 if lambda <= 100 else 10. A HashMap<Float,Float>:
        duplicate lambdas collapse."""
        out: Dict[np.float32, np.float32] = {}
        for j, l in enumerate(self.lambdas):
            lf = np.float32(l)
            if self.rhos is not None:
                out[lf] = np.float32(self.rhos[j])
            else:
                out[lf] = np.float32(1.0) if lf <= 100 else np.float32(10.0)
        return out

    def sorted_lambda_rho(self):
        lr = self.lambda_rho()
        lam = sorted(lr.keys())                                           # :636-638
        return [float(l) for l in lam], [float(lr[l]) for l in lam]


# ------------------------------------------------------------------------------- driver loop
@dataclass
class IterationRecord:
    iteration: int
    liblinear_epsilon: float
    rho_adapt_rate: float
    maxdiff: float
    mindiff: float
    stats: Any = None
    test_loglik: Optional[Dict[str, float]] = None      # lambda key -> mean weighted test log-likelihood


class AdmmTrain:
    """The outer loop of jobs/RegressionAdmmTrain.java:278-501 around an engine.

    Engine protocol: ``solve_local(eps, rate)``, ``consensus_finish() -> obj with maxdiff/mindiff``,
    ``z() -> (Z double, z float32)``; for a sharded run additionally ``consensus_tensor()`` returning a
    torch tensor that aliases the [xbar | ubar] buffer, summed in place over ranks by ``all_reduce``.
    ``all_reduce(tensor)`` must have COMPLETED when it returns (with torch.distributed on a GPU: issue the collective,
    then synchronize the stream it was ordered on) -- the engine's next kernels run on the engine's own stream.
    """

    def __init__(self, config: AdmmConfig, engine, all_reduce=None):
        self.cfg = config
        self.engine = engine
        self.all_reduce = all_reduce
        self.history: List[IterationRecord] = []
        self.test_n: Optional[float] = None             # sum of test weights; set by attach_test_rows
        self.best_test_loglik = np.float32(-9999999)    # MutableFloat(-9999999), :234
        self.best_model = None                          # (iteration, lambda key, float32 z) like best-model/best-iteration-i.avro

    def attach_test_rows(self, rows) -> None:
        """test.path handling (:203-232): upload the first file's rows once; loglik is then drawn every iteration."""
        self.engine.set_test_data(rows.row_ptr, rows.global_idx, rows.val, rows.response, rows.weight, rows.offset)
        self.test_n = float(rows.n)

    def _update_loglik_best_model(self, niter: int) -> Dict[str, float]:
        """updateLogLikBestModel (:812-845): loglik[lambda] = sum / n; best model when loglik > best (float compare), niter > 0."""
        lam, _ = self.cfg.sorted_lambda_rho()
        sums = self.engine.test_loglik_sums()
        out: Dict[str, float] = {}
        for li, l in enumerate(lam):
            key = java_float_to_string(l)
            ll = float(sums[li]) / self.test_n
            out[key] = ll
            if ll > float(self.best_test_loglik) and niter > 0:
                self.best_model = (niter, key, self.engine.z()[1][li].copy())
                self.best_test_loglik = np.float32(ll)
        return out

    def rho_adapt_rate(self, i: int) -> float:
        """conf RHO_ADAPT_RATE for iteration i (:313-317,323-327) (the JobConf is rebuilt every iteration)."""
        c = np.float32(self.cfg.rho_adapt_coefficient)
        if i > 1 and c > 0:
            x = -(np.float32(i - 1) * c)                                  # int*float in float, :325
            return float(np.float32(math.exp(float(x))))
        return 1.0

    def mean_model_init(self) -> None:
        """Initialize z by the mean model (:236-276): the RegressionNaiveTrain job on the same partitions
        (liblinear.epsilon from the job file, else 0.01, :246-249; prior.mean) and z = meanModel(...); with
        test.loglik.per.iter the loglik of that z is drawn as iteration 0 (:271-274; never the best model)."""
        p = self.cfg.extras
        eps = float_string_roundtrip(np.float32(p.get("liblinear.epsilon", 0.01)))
        prior_mean = float(np.float32(p.get("prior.mean", 0.0)))
        self.init_stats = self.engine.naive_solve_local(eps, prior_mean)
        if self.all_reduce is not None:
            self.all_reduce(self.engine.consensus_tensor())
        self.engine.naive_finish()
        if self.test_n is not None:
            self.init_test_loglik = self._update_loglik_best_model(0)

    def run(self, callback=None) -> List[IterationRecord]:
        cfg = self.cfg
        boost = cfg.initialize_boost_rate > 0 and cfg.regularizer == 2
        if boost:
            self.mean_model_init()
        mindiff = 99999999.0
        e = np.float32(0.01)                                              # :279
        for i in range(1, cfg.num_iters + 1):
            # RHO_ADAPT_RATE lives in the per-iteration JobConf (:287-291): the boost rate at i == 1 (:313-317),
            # exp(-(i-1)c) from i == 2 on when rho.adapt.coefficient > 0 (:323-327), else the reducer default 1
            rate = 1.0
            if i == 1 and boost:
                rate = float(np.float32(cfg.initialize_boost_rate))
            if i > 1 and cfg.rho_adapt_coefficient > 0:
                rate = self.rho_adapt_rate(i)
            if i > 1 and mindiff < 0.001 and not cfg.aggressive_liblinear_epsilon_decay:
                e = np.float32(e / np.float32(10))                        # :338-341 float division
            elif cfg.aggressive_liblinear_epsilon_decay and i > 5:
                e = np.float32(e / np.float32(10))                        # :342-345
            eps = float_string_roundtrip(e)
            st = self.engine.solve_local(eps, rate)
            if self.all_reduce is not None:
                self.all_reduce(self.engine.consensus_tensor())
            fin = self.engine.consensus_finish()
            maxdiff, mindiff = float(fin.maxdiff), float(fin.mindiff)
            rec = IterationRecord(i, eps, rate, maxdiff, mindiff, st)
            if self.test_n is not None:
                rec.test_loglik = self._update_loglik_best_model(i)
            self.history.append(rec)
            if callback is not None:
                callback(rec)
            if maxdiff < cfg.epsilon and float(e) <= 0.00001:             # :493-496
                break
        return self.history

    def final_models(self) -> Dict[str, np.ndarray]:
        """key String.valueOf(lambda) -> float32 coefficient vector (global index, intercept last)."""
        lam, _ = self.cfg.sorted_lambda_rho()
        _, z32 = self.engine.z()
        return {java_float_to_string(l): z32[i] for i, l in enumerate(lam)}


# ------------------------------------------------------------------------------- model files
def model_to_avro(vec: np.ndarray, feature_names: Sequence[str], skip_zero: bool = False) -> List[Dict[str, Any]]:
    """models/LinearModel.java:697-720 ``toAvro``: intercept record first, then name/term split on U+0001."""
    out = [{"name": INTERCEPT_NAME, "term": "", "value": float(np.float32(vec[-1]))}]
    for j, key in enumerate(feature_names):
        v = np.float32(vec[j])
        if skip_zero and v == 0:
            continue
        tok = key.split(TERM_SEP)
        out.append({"name": tok[0], "term": tok[1] if len(tok) > 1 else "", "value": float(v)})
    return out


def write_linear_models(path: str, models: Dict[str, np.ndarray], feature_names: Sequence[str]) -> None:
    """utils/LinearModelUtils.java:39-53 (``final-model/part-r-00000.avro``, ``iter-i/init-value`` ...)."""
    recs = [{"key": k, "model": model_to_avro(v, feature_names)} for k, v in models.items()]
    avro_io.write_container(path, avro_io.LINEAR_MODEL_SCHEMA, recs)


def sample_test_loglik_records(history: Sequence[IterationRecord]) -> List[Dict[str, Any]]:
    """sample-test-loglik/iteration-i.avro records (avro/SampleTestLoglik.avsc; testLoglik stored as float, :828)."""
    out = []
    for h in history:
        for key, ll in (h.test_loglik or {}).items():
            out.append({"lambda": key, "iter": h.iteration, "testLoglik": float(np.float32(ll))})
    return out


def read_linear_models(path: str, feature_names: Sequence[str]) -> Dict[str, np.ndarray]:
    """models/LinearModel.java:112-156 on every record; returns dense float32 vectors (intercept last)."""
    index = {k: j for j, k in enumerate(feature_names)}
    out: Dict[str, np.ndarray] = {}
    for rec in avro_io.read_records(path):
        v = np.zeros(len(feature_names) + 1, np.float32)
        for f in rec["model"]:
            name = f["name"] if f["term"] == "" else f["name"] + TERM_SEP + f["term"]
            if name == INTERCEPT_NAME:
                v[-1] = np.float32(f["value"])
            elif name in index:
                v[index[name]] = np.float32(f["value"])
        out[rec["key"]] = v
    return out


# ------------------------------------------------------------------------------- RegressionTest (scoring job)
def read_model_file(path: str) -> Dict[str, Tuple[List[str], np.ndarray]]:
    """ReadLinearModelConsumer over a model file or directory: key -> (feature keys in file order, float32 vector with the
    intercept last). A later record with the same key replaces an earlier one (HashMap.put)."""
    out: Dict[str, Tuple[List[str], np.ndarray]] = {}
    for rec in avro_io.read_records(path):
        names: List[str] = []
        vals: List[float] = []
        icpt = np.float32(0)
        for f in rec["model"]:
            name = f["name"] if f["term"] in ("", None) else f["name"] + TERM_SEP + f["term"]
            if name == INTERCEPT_NAME:
                icpt = np.float32(f["value"])
            else:
                names.append(name)
                vals.append(f["value"])
        out[str(rec["key"])] = (names, np.asarray(vals + [icpt], np.float32))
    return out


def test_output_schema(input_schema: Any) -> Dict[str, Any]:
    """jobs/RegressionTest.java:201-232: record AdmmTestOutput = the input fields (Util.removeUnion of the top level) +
    `pred` float."""
    sch = input_schema
    if isinstance(sch, list):                                             # top-level union: the record branch
        sch = next(s for s in sch if isinstance(s, dict) and s.get("type") == "record")
    fields = [{"name": f["name"], "type": f["type"], **({"doc": f["doc"]} if "doc" in f else {})} for f in sch["fields"]]
    fields.append({"name": "pred", "type": "float", "doc": ""})
    return {"type": "record", "name": "AdmmTestOutput", "namespace": "com.linkedin.lab.regression.avro",
            "doc": "Test output for AdmmTest", "fields": fields}


test_output_schema.__test__ = False


def regression_test(props: Dict[str, str], scorer) -> List[str]:
    """jobs/RegressionTest.java:64-112 for local files: for every lambda of the job, score input.paths with the
    final-model record of that lambda (key String.valueOf(Float.parseFloat(lambda))) and write
    output.base.path/lambda-<lambda as written>/part-r-00000.avro = input fields + pred, ordered by pred like the
    shuffle on the Float key orders them (ties keep input order here; Hadoop leaves them unspecified); then the same with
    model.base.path/best-model if it exists (the first model of the directory). `scorer.score_rows(model32, row_ptr,
    global_idx, val, offset)` is HipScorer in the product. Returns the files written."""
    from . import dataset
    in_path = props.get("input.paths", "")
    if in_path == "":
        return []                                                          # "test.input.paths is empty!" :109-111
    out_base, model_base = props["output.base.path"], props["model.base.path"]
    binary = _get_bool(props, "binary.feature", False)
    files = sorted(os.path.join(in_path, n) for n in os.listdir(in_path) if n.endswith(".avro")) if os.path.isdir(in_path) else [in_path]
    schema, _ = avro_io.read_container(files[0])
    records = [r for f in files for r in avro_io.read_container(f)[1]]
    out_schema = test_output_schema(schema)
    written: List[str] = []

    def run(model_names: List[str], model: np.ndarray, out_dir: str) -> None:
        rows = dataset.build_test_rows(records, model_names, binary, max_rows=len(records) + 1)
        pred = scorer.score_rows(model, rows.row_ptr, rows.global_idx, rows.val, rows.offset)
        order = np.argsort(pred, kind="stable")
        recs = [dict(records[i], pred=float(pred[i])) for i in order]
        path = os.path.join(out_dir, "part-r-00000.avro")
        avro_io.write_container(path, out_schema, recs)
        written.append(path)

    models = read_model_file(os.path.join(model_base, "final-model"))
    for lam in [s for s in props["lambda"].split(",") if s.strip() != ""]:
        key = java_float_to_string(np.float32(lam))
        if key not in models:
            raise KeyError("no model for lambda %s in %s/final-model" % (key, model_base))    # the reference NPEs in map()
        run(models[key][0], models[key][1], os.path.join(out_base, "lambda-" + lam))
    best = os.path.join(model_base, "best-model")
    if os.path.isdir(best) and any(n.endswith(".avro") for n in os.listdir(best)):
        bm = read_model_file(best)
        names, vec = next(iter(bm.values()))
        run(names, vec, os.path.join(out_base, "best-model"))
    return written
