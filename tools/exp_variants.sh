# usage: bash tools/exp_variants.sh lib1.so lib2.so ...   (bench each variant, plus the default build)
echo "default"; timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null > /tmp/b.log; python tools/show_bench.py /tmp/b.log
for lib in "$@"; do
  echo "$lib"; SOTS_LIB_PATH=$lib timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null > /tmp/b.log; python tools/show_bench.py /tmp/b.log
done
