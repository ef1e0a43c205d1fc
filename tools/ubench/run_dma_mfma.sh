cd $GRAFT_REPO_ROOT/tools/ubench
O=../../gpurun_out/dma_mfma4.txt; : > $O
for p in 640 1280 2560 5120 5760 11520 23040 704 1408 2688; do
  ./dma_mfma 0 4 0 1000 8 256 1 2 0 $p >> $O
done
./dma_mfma 2 4 0 1000 8 256 1 2 0 1280 >> $O
for p in 640 1280 2560 5760 11520; do
  ./dma_mfma 0 4 25 1000 8 256 1 2 0 $p >> $O
done
./dma_mfma 2 4 25 1000 8 256 1 2 0 1280 >> $O
cat $O | cut -c1-150
