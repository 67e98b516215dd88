# sub-frames x frames in flight for the per-rank shares of the deep scenes (tools/time_bands.py); run on the GPU box
set -e
for sc in "s1m 5" "s100k 5"; do
for ch in 3,1 3,2 2,2 2,3 1,3 1,4 2,4 1,6 4,2; do
echo "== $sc chains $ch"
RT_TIME_BANDS_CHAINS=$ch timeout -k 10 120 python -u tools/time_bands.py $sc 2>/dev/null
done; done
