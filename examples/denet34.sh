#!/bin/bash
# DeNet-34 (std | skip) on Pascal VOC or MSCOCO with this build's drivers: the command sequence of the reference's
# papers/dss/denet34.sh (model-modify x2 -> model-train-multi -> model-predict) with the same flags.
#   examples/denet34.sh <std|skip> <voc2007|mscoco> <dataset dir> [pretrained resnet34.mdl.gz]
# Without a pretrained classifier the ResNet-34 backbone starts from random weights (there is no network in this
# environment to fetch models/imagenet/resnet34.mdl.gz).
set -e
MODEL_VAR=${1:?model variant: std | skip}
DATASET=${2:?dataset: voc2007 | mscoco}
INPUT_DIR=${3:?dataset directory}
BASE_MODEL=$4
DIR="$( cd "$( dirname "${BASH_SOURCE[0]}" )" && pwd )/.."
BIN=$DIR/bin

if [[ $MODEL_VAR == "std" ]]; then
    MODEL_DESC="PI[2] C.B[256,3] BNA PI[2] C.B[128,3] BNA DNC[96,100] DNS[7,24,0.01,0.1] C.B[1536,1] BNA C.B[1024,1] BNA C.B[768,1] BNA C.B[512,1] BNA DND[0.5,1,1]"
else
    MODEL_DESC="PI[2] C[256,3] SKIP[1] BNA PI[2] C[128,3] SKIP[0] BNA DNC[96,100] DNS[7,24,0.01,0.1] C[1536,1] BNA C.B[1024,1] BNA C.B[768,1] BNA C.B[512,1] BNA DND[0.5,1,1]"
fi
IMAGE_LOADER="images_per_subset=1280,scale=512,crop=512,augment_photo,crop_mode=denet,scale_mode=large"
EPOCHS=${EPOCHS:-90}
BATCH=${BATCH:-32}
TRAIN_PARAM="--solver nesterov --epochs $EPOCHS --batch-size $BATCH --batch-size-factor 2 --learn-rate 0.1 --learn-momentum 0.9 --learn-anneal 0.1 --learn-anneal-epochs 30 60 --learn-decay 0.0001"
if [[ $DATASET == "voc2007" ]]; then
    DATA_TYPE=voc; TRAIN_DATA=2007-trainval,2012-trainval; TEST_DATA=2007-test; CLASS_NUM=20
else
    DATA_TYPE=mscoco; TRAIN_DATA=2014-train,2014-val; TEST_DATA=2015-test; CLASS_NUM=80
fi
OUTPUT_DIR=./denet34-$DATASET-$MODEL_VAR
mkdir -p $OUTPUT_DIR && cd $OUTPUT_DIR

if [ -z "$BASE_MODEL" ]; then
    BASE_MODEL=./resnet34-random.mdl.gz
    PYTHONPATH=$DIR python - <<PY
from denet_amd.model import model_cnn, zoo
model_cnn.save_to_file(zoo.resnet34(32), "$BASE_MODEL")
PY
fi

if [ ! -f ./initial.mdl.gz ]; then
    if [[ $MODEL_VAR == "skip" ]]; then
        $BIN/model-modify --input $BASE_MODEL --output initial_skipsrc.mdl.gz --modify-bn 1 0.9 1e-5 --convert-bn-relu --use-cudnn-pool --class-num $CLASS_NUM --image-size 512 512 --layer-remove 3 --layer-insert "11:SKIPSRC.X[0]" "18:SKIPSRC.X[1]"
        $BIN/model-modify --input initial_skipsrc.mdl.gz --output initial.mdl.gz --layer-append $MODEL_DESC
    else
        $BIN/model-modify --input $BASE_MODEL --output initial.mdl.gz --modify-bn 1 0.9 1e-5 --convert-bn-relu --use-cudnn-pool --class-num $CLASS_NUM --image-size 512 512 --layer-remove 3 --layer-append $MODEL_DESC
    fi
fi

# one rank per visible GPU (GPUS=N to restrict); --device-render: crops / resampling / jitter on the GPU
$BIN/model-train-multi $TRAIN_PARAM --thread-num 8 --seed 1 --device-render --border-mode half --model initial.mdl.gz \
    --train "$INPUT_DIR" --extension $DATA_TYPE,$TRAIN_DATA,$IMAGE_LOADER --output-prefix ./model

$BIN/model-predict --batch-size $BATCH --thread-num 8 --predict-mode detect,$DATA_TYPE --device-render \
    --model ./model_epoch$(printf "%03d" $((EPOCHS-1)))_final.mdl.gz --input "$INPUT_DIR" --extension $DATA_TYPE,$TEST_DATA,images_per_subset=1280,scale=512,crop=512,scale_mode=large \
    --results ./predict/results --params "prThreshold=0.01,nmsThreshold=0.5"
