cd $GRAFT_REPO_ROOT
echo "== normal"; STEPS=10 bash tools/bench_all.sh fixed_base
cp babyjubjub-rs_amd/csrc/libbjj_hip.so /tmp/keep.so; cp tools/libbjj_noinv_experiment.so babyjubjub-rs_amd/csrc/libbjj_hip.so
echo "== no inversion (wrong results, timing only)"; STEPS=10 bash tools/bench_all.sh fixed_base
cp /tmp/keep.so babyjubjub-rs_amd/csrc/libbjj_hip.so
