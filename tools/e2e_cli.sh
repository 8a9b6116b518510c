#!/bin/bash
# Run ON THE GPU BOX: end-to-end wall clock of the native CLI on synthetic one-hot input (configs[2] generator),
# raw avro -> Prepare semantics -> indexing -> upload -> ADMM -> final-model. Usage: tools/e2e_cli.sh <rows> <partitions> [iters]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
ROWS=${1:-2000000}; PARTS=${2:-64}; ITERS=${3:-20}
g++ -O2 -std=c++17 -I $R/ml-ease_amd/host $R/tools/gen_onehot_avro.cpp $R/ml-ease_amd/host/avro_io.o -lz -o /tmp/gen_onehot_avro || exit 1
rm -rf /tmp/oh /tmp/oh_out
/tmp/gen_onehot_avro /tmp/oh $ROWS 16
du -sh /tmp/oh
cat > /tmp/oh.job <<EOJ
input.paths=/tmp/oh
output.base.path=/tmp/oh_out
num.blocks=$PARTS
lambda=1.0
num.iters=$ITERS
regularizer=2
binary.feature=${BINARY:-true}
EOJ
$R/ml-ease_amd/host/mlease_admm_train /tmp/oh.job 2>&1 | grep -v "iteration [0-9]*:" | tail -8
ls -la /tmp/oh_out/final-model/
