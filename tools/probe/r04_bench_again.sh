cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=gpurun_out/r04/profiles; mkdir -p $P
timeout 900 python bench.py > $P/r04_bench.json 2> gpurun_out/r04/bench.err; echo "bench rc=$?"
(python tests/tools/model_bench.py f16 split f32 2>&1 | grep model; python tests/tools/model_bench.py --hetero f16 split 2>&1 | grep model;
 python tests/tools/encoder_bench.py 2>&1 | grep PointPillar;
 python tests/tools/camera_bench.py f16 split f32 2>&1 | grep Cvt) > $P/r04_model.txt; echo "model rc=$?"
cut -c1-200 $P/r04_bench.json; cat $P/r04_model.txt
