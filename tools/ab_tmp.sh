for c in configs1 configs0; do for i in 1 2; do
 GMVAE_HIP_LIB=$PWD/build_ab/libA.so python tools/step_time.py $c 1.0 2>&1 | tail -1 | cut -c1-70
 python tools/step_time.py $c 1.0 2>&1 | tail -1 | cut -c1-70
done; done
