for i in 1 2 3; do
 GMVAE_NO_FLSPLIT=1 python tools/step_time.py configs2 1.0 2>&1 | tail -1 | cut -c1-200
 python tools/step_time.py configs2 1.0 2>&1 | tail -1 | cut -c1-260
done
