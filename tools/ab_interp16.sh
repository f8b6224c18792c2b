#!/bin/bash
# A/B of interp_pool_f16x3_kernel build variants (python -m ppsurf_amd.build --variant ...), timed with tools/time_decoder_stages.py:
#   base      product build (IH_NT=512, IH_OB=2: one 8-wave workgroup per CU, 32 KiB chunks)
#   pf2       A fragments of the split-precision layers requested two k-steps ahead (ih256 = two decoupled 4-wave workgroups and ihob4 = 64 KiB chunks
#             were measured in round 3: 2.74 and 2.67 ms against 2.65 -- neither the barrier count nor the phase lock is what bounds the kernel)
#   *nosm     softmax + pooling ablated, *nosmg: and the gather -> what the MFMA phase alone costs
# build here (CPU container):  tools/ab_interp16.sh build        run on the GPU box:  tools/ab_interp16.sh
VARIANTS="burst:-DIH_SPREAD=0"
if [ "$1" = build ]; then
  for v in $VARIANTS; do name=${v%%:*}; flags=${v#*:}; python -m ppsurf_amd.build --variant $name ${flags//,/ } > /dev/null || exit 1; echo built $name; done
  exit 0
fi
echo base; python tools/time_decoder_stages.py 30
for v in $VARIANTS; do name=${v%%:*}; echo $name; PPS_LIB_VARIANT=$name python tools/time_decoder_stages.py 30 ; done
