# frames in flight x frames per launch chain, longer sequences; run on the GPU box
set -e
for fl in 1 2 3; do for v in 4 8; do
echo "== s1m seq 48 flight $fl stack $v"
RT_TIME_BANDS_SEQ=48 RT_TIME_BANDS_PARTS=1,8 RT_TIME_BANDS_CHAINS=1,$fl RT_TIME_BANDS_STACK=$v timeout -k 10 150 python -u tools/time_bands.py s1m 5 2>/dev/null
done; done
