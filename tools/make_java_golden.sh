#!/bin/bash
# make_java_golden.sh -- pin the oracle (and the HIP path) to the REAL reference: run linkedin/ml-ease's own Java on
# examples/sample-data.avro and keep its model files as fixtures under tests/golden/java/.
#
# Needs what this repository's build image does NOT have: a JDK (>= 6), maven, and maven's dependencies (network or a
# populated ~/.m2: hadoop-core 1.2.1, avro 1.7.6, ... -- pom.xml:36-38,99-120). Run it wherever those exist:
#
#     tools/make_java_golden.sh [/path/to/ml-ease checkout]        (default /root/reference; never written to: built in a copy)
#
# then `python -m pytest tests/test_java_golden.py` compares oracle/admm_oracle.c -- and, with -m gpu, the HIP library -- with
# the files it leaves (tests skip while tests/golden/java/ holds no model). Commit the fixtures: they are a few KB.
#
# Two jobs, both through the reference's own entry point jobs/Regression.java:88 (RegressionPrepare -> RegressionAdmmTrain)
# with is.local=true (mapred/AbstractAvroJob.java:260-267: local job tracker, file:/// file system):
#   blocks1  num.blocks=1, lambda=1.0, 20 iterations. One reduce key (0), so Hadoop 1.2.1's LocalJobRunner -- which forces a single
#            reduce task -- satisfies AdmmPartitioner (jobs/RegressionAdmmTrain.java:575-590 throws for a key >= numPartitions).
#            The random partition key (jobs/RegressionPrepare.java:112) is floor(random * 1) = 0: deterministic without map.key.
#            Pins R1-R10, R12-R15 of SURVEY 8(a) on real data (the mean over one block is the identity).
#   blocks8  num.blocks=8, map.key=pkey (a copy of the sample with pkey = row index % 8: BASELINE configs[0] as
#            tests/golden/c1_golden.npz has it). Needs a runner that grants 8 reduce tasks (a pseudo-distributed Hadoop 1.x, or
#            a Hadoop >= 2 LocalJobRunner); under the plain 1.2.1 local runner this job FAILS in the partitioner and is skipped.
set -u
REF=${1:-/root/reference}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/tests/golden/java
command -v java >/dev/null && command -v mvn >/dev/null || { echo "needs java and mvn on PATH (not in the build image)"; exit 2; }
[ -f "$REF/pom.xml" ] || { echo "no ml-ease checkout at $REF"; exit 2; }
WORK=$(mktemp -d /tmp/mlease_java_golden.XXXXXX)
cp -r "$REF" "$WORK/src" && cd "$WORK/src" || exit 1
mvn -q -DskipTests clean package || { echo "mvn package failed (dependencies reachable?)"; exit 1; }
JAR=$(ls target/*-jar-with-dependencies.jar | head -1)
[ -f "$JAR" ] || { echo "no jar-with-dependencies under target/"; exit 1; }
mkdir -p "$WORK/in1" "$WORK/in8"
cp examples/sample-data.avro "$WORK/in1/part-00000.avro"
# the keyed copy for the 8-block job (this repository's avro writer; the records are otherwise unchanged)
python3 - "$ROOT" "$WORK" <<'PY'
import sys
root, work = sys.argv[1], sys.argv[2]
sys.path.insert(0, root)
import mlease_amd  # noqa: F401
from mlease_amd import avro_io
schema, it = avro_io.read_container(work + "/in1/part-00000.avro")
recs = list(it)
for i, r in enumerate(recs):
    r["pkey"] = str(i % 8)
schema["fields"].append({"name": "pkey", "type": "string"})
avro_io.write_container(work + "/in8/part-00000.avro", schema, recs, codec="deflate")
PY
run_job() {   # name num.blocks extra-line
    local name=$1 nb=$2 extra=$3
    cat > "$WORK/$name.job" <<JOB
input.paths=$WORK/in$nb
output.base.path=$WORK/out_$name
num.blocks=$nb
lambda=1.0
num.iters=20
regularizer=2
is.local=true
force.output.overwrite=true
remove.tmp.dir=false
$extra
JOB
    java -Xmx2g -cp "$JAR" com.linkedin.mlease.regression.jobs.Regression "$WORK/$name.job" > "$WORK/$name.log" 2>&1 || {
        echo "$name: the reference job failed (see $WORK/$name.log)"; tail -5 "$WORK/$name.log"; return 1; }
    mkdir -p "$OUT/$name"
    cp "$WORK/$name.job" "$OUT/$name/job.properties"
    for i in $(seq 1 20); do
        for d in model u init-value; do
            [ -d "$WORK/out_$name/iter-$i/$d" ] || continue
            mkdir -p "$OUT/$name/iter-$i/$d"
            cp "$WORK/out_$name/iter-$i/$d"/*.avro "$OUT/$name/iter-$i/$d/"
        done
    done
    mkdir -p "$OUT/$name/final-model" && cp "$WORK/out_$name/final-model"/*.avro "$OUT/$name/final-model/"
    cp "$WORK/out_$name/lambda-rho" "$OUT/$name/" 2>/dev/null || cp -r "$WORK/out_$name/lambda-rho"* "$OUT/$name/" 2>/dev/null
    echo "$name: fixtures under $OUT/$name"
}
run_job blocks1 1 ""
run_job blocks8 8 "map.key=pkey" || echo "blocks8 skipped: this Hadoop's local runner has a single reduce task (see the header)"
echo "now: python -m pytest tests/test_java_golden.py -q   (and -m gpu on an MI355X)"
