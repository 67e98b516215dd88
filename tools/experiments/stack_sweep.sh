# sweep of frames per launch chain (RT_TIME_BANDS_STACK) for the per-rank shares of the deep scenes; run on the GPU box
set -e
for sc in "s1m 5" "s100k 5"; do
for v in 1 2 4 8; do
echo "== $sc stack $v"
RT_TIME_BANDS_STACK=$v timeout -k 10 150 python -u tools/time_bands.py $sc 2>/dev/null
done; done
