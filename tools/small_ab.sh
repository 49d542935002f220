# same-box A/B of two library builds over the small-batch regime: LIBS="libdfm_r03c libdfmdock_amd" bash tools/small_ab.sh
cd $GRAFT_REPO_ROOT
for lib in ${LIBS:-libdfm_r03c libdfmdock_amd libdfm_r03c libdfmdock_amd}; do
  DFM_LIB=$GRAFT_REPO_ROOT/dfmdock_amd/$lib.so python tools/small_batch.py 2>&1 | grep "B="
done
