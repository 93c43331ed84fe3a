# usage: bash tools/mkvariant.sh NAME -DFLAG[=V] ...   -> variants/NAME.so (the kernels rebuilt with the flags, the other objects as built)
set -e
cd "$(dirname "$0")/../survival_of_the_synthesis-gpu_accelerated_frequency_modulation_parameter_matcher_amd"
name=$1; shift
mkdir -p ../variants build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -c -o build/variant_$name.o csrc/sots_kernels.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../variants/$name.so build/variant_$name.o build/sots_capi.o build/sots_group.o build/sots_host_math.o -ldl -lpthread
echo variants/$name.so
