# diagnostic / A-B builds of temporal_block640.hip: usage build_variants.sh name1:"-DX=1 -DY=2" name2:"..."  ->  synfmc_amd/lib/knock/libfmc_hip_<name>.so
set -e
cd "$(dirname "$0")/../../../synfmc_amd/csrc"
mkdir -p ../lib/knock
SRC=${SRC:-temporal_block640}
OBJS=$(ls ../lib/obj/*.o | grep -v "/$SRC.o")
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DFMC_GELU_EXACT=0 $defs -c $SRC.hip -o ../lib/knock/${SRC}_$name.o &
done
wait
for spec in "$@"; do
  name=${spec%%:*}
  hipcc --offload-arch=gfx950 -shared -fPIC $OBJS ../lib/knock/${SRC}_$name.o -lhipblaslt -o ../lib/knock/libfmc_hip_$name.so
  rm ../lib/knock/${SRC}_$name.o
done
ls ../lib/knock
