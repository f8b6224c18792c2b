#!/bin/bash
# A/B of interp_pool_f16x3_kernel build variants (python -m ppsurf_amd.build --variant ...), timed with tools/time_decoder_stages.py:
#   base      product build (IH_NT=512, IH_OB=2: one 8-wave workgroup per CU, 32 KiB chunks)
#   ih256     two decoupled 4-wave workgroups per CU (the weight stream doubles, the VALU phases of one overlap the MFMA phase of the other)
#   ihob4     64 KiB chunks (9 instead of 18 barriers per pass)
#   *nosm     softmax + pooling ablated, *nosmg: and the gather -> what the MFMA phase alone costs
# build here (CPU container):  tools/ab_interp16.sh build        run on the GPU box:  tools/ab_interp16.sh
VARIANTS="ih256:-DIH_NT=256 ihob4:-DIH_OB=4 ihnosm:-DPPS_ABL_IH_NOSOFTMAX ihnosmg:-DPPS_ABL_IH_NOSOFTMAX,-DPPS_ABL_IH_NOGATHER ih256nosm:-DIH_NT=256,-DPPS_ABL_IH_NOSOFTMAX ih256nosmg:-DIH_NT=256,-DPPS_ABL_IH_NOSOFTMAX,-DPPS_ABL_IH_NOGATHER"
if [ "$1" = build ]; then
  for v in $VARIANTS; do name=${v%%:*}; flags=${v#*:}; python -m ppsurf_amd.build --variant $name ${flags//,/ } > /dev/null || exit 1; echo built $name; done
  exit 0
fi
echo base; python tools/time_decoder_stages.py 30 | grep f16x3
for v in $VARIANTS; do name=${v%%:*}; echo $name; PPS_LIB_VARIANT=$name python tools/time_decoder_stages.py 30 | grep f16x3; done
