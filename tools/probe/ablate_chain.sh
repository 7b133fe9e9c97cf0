# builds ablation variants of the chain kernels into tools/probe/lib_<name>.so (run where hipcc is; the objects of
# the other sources must have been built by `make` first)
cd /root/repo/hm-vit_amd/csrc
for v in BASE "FFN_NO_STORES" "NO_DMA" "FFN_NO_MFMA" "FFN_NO_GELU" "FFN_NO_MFMA -DNO_DMA -DFFN_NO_GELU" "FFN_NO_MFMA -DNO_DMA -DFFN_NO_GELU -DFFN_NO_STORES"; do
  name=$(echo $v | sed 's/ -D/+/g')
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -D$v -c chain.hip -o /tmp/t/chain_ab.o 2>/dev/null
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 tok.o gemm.o attn.o /tmp/t/chain_ab.o split.o enc.o capi.o -o /root/repo/tools/probe/lib_$name.so
  echo built $name
done
