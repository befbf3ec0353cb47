cd $GRAFT_REPO_ROOT
for i in 1 2; do
echo "== current"; python tools/wino_conv_bench.py abl
echo "== base (first commit of the kernel, non-persistent)"; GD_NN_LIB=$PWD/ablate/libgd_nn_base.so python tools/wino_conv_bench.py abl
done
