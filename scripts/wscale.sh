timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for w in 16 13 10 8; do echo "waves cap $w"; RAPID_TALLY_WAVES=$w timeout 120 python scripts/ablate.py 2>&1 | head -3 | cut -c1-60; done
