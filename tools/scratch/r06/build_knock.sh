# diagnostic builds of geglu_direct_kernel with parts knocked out (G6_KNOCK bits: 1 = no weight loads, 2 = no LDS fragment reads, 4 = no epilogue)
set -e
cd "$(dirname "$0")/../../../synfmc_amd/csrc"
mkdir -p ../lib/knock
OBJS=$(ls ../lib/obj/*.o | grep -v temporal_block640.o)
for n in 1 2 3 4 5 6 7; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DFMC_GELU_EXACT=0 -DG6_KNOCK=$n -c temporal_block640.hip -o ../lib/knock/tb640_k$n.o &
done
wait
for n in 1 2 3 4 5 6 7; do
  hipcc --offload-arch=gfx950 -shared -fPIC $OBJS ../lib/knock/tb640_k$n.o -lhipblaslt -o ../lib/knock/libfmc_hip_k$n.so
  rm ../lib/knock/tb640_k$n.o
done
ls -la ../lib/knock
