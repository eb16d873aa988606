for i in 1 2; do
 GMVAE_SK_NO_FORK=1 python tools/step_time.py run_train 1.0 2>&1 | tail -1 | cut -c1-60
 python tools/step_time.py run_train 1.0 2>&1 | tail -1 | cut -c1-60
done
