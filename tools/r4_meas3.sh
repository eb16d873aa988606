cd $GRAFT_REPO_ROOT
for m in "vae_gmp 256 64 10" "vae 100 2 1"; do
  echo "=== $m"; GMVAE_STAMPS=1 python tools/stamps.py $m 2>&1 | tail -7
  GMVAE_STAMPS=2 python tools/stamps.py $m 2>&1 | tail -4
  GMVAE_STAMPS=4 python tools/stamps.py $m 2>&1 | tail -10
done
