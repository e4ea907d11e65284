timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
for lib in librapid_mi355x.so librapid_mi355x_q2r10.so librapid_mi355x_q4r10.so librapid_mi355x_q2r7.so; do
  echo "== $lib"; RAPID_MI355X_LIB=$PWD/rapid_amd/$lib timeout 120 python scripts/ablate.py 2>&1 | head -3 | cut -c1-48
done
