for i in 1 2; do
 GMVAE_HIP_LIB=$PWD/build_ab/libPrev.so python tools/step_time.py run_train 1.0 2>&1 | tail -1 | cut -c1-60
 python tools/step_time.py run_train 1.0 2>&1 | tail -1 | cut -c1-60
 GMVAE_SK_DW_RR=1 python tools/step_time.py run_train 1.0 2>&1 | tail -1 | cut -c1-60
done
