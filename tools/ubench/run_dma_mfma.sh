cd $GRAFT_REPO_ROOT/tools/ubench
O=../../gpurun_out/dma_mfma3.txt; : > $O
for sh in 0 2; do
for w in 4 8; do
  ./dma_mfma $sh 4 0 2000 $w 256 1 2 0 >> $O
  ./dma_mfma $sh 0 25 2000 $w 256 1 2 0 >> $O
  ./dma_mfma $sh 4 25 2000 $w 256 1 2 0 >> $O
  ./dma_mfma $sh 4 25 2000 $w 256 1 2 1 >> $O
  ./dma_mfma $sh 4 25 2000 $w 256 1 2 2 >> $O
  ./dma_mfma $sh 4 50 2000 $w 256 1 2 0 >> $O
  ./dma_mfma $sh 4 50 2000 $w 256 1 2 1 >> $O
  ./dma_mfma $sh 4 50 2000 $w 256 1 2 2 >> $O
done
done
cat $O | cut -c1-170
