# usage: tools/sweep.sh "<time_render args>"  -- runs tools/time_render.py with every build/librt_*.so (development aid)
for f in build/librt_*.so; do echo "== $f"; RT_HOTPATH_LIB=$f timeout -k 10 60 python tools/time_render.py $1 || exit 1; done
