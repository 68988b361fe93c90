#!/bin/bash
# runs examples/denet34.sh end to end on the synthetic VOC tree of the test-suite (1 epoch, batch 2): model-modify x2,
# model-train-multi (1 rank, device-rendered data), model-predict with the VOC writers
set -e
ROOT="$( cd "$( dirname "${BASH_SOURCE[0]}" )" && pwd )/.."
TMP=$(mktemp -d)
PYTHONPATH=$ROOT:$ROOT/tests/golden python -c "import dataset_scenarios as S; S.build_dataset('$TMP/data')"
cd $TMP
EPOCHS=1 BATCH=2 GPUS=1 MASTER_PORT=29541 bash $ROOT/examples/denet34.sh skip voc2007 $TMP/data/voc
ls $TMP/denet34-voc2007-skip $TMP/denet34-voc2007-skip/predict | head -30
