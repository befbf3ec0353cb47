cd $GRAFT_REPO_ROOT
echo "product"; python tools/wino_conv_bench.py abl
for n in ${WINO_ABL:-1 2 3 4 5 6 7}; do echo "ablate $n"; GD_NN_LIB=$PWD/ablate/libgd_nn_w$n.so python tools/wino_conv_bench.py abl; done
