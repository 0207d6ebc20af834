#!/bin/bash
# The whole -m gpu suite twice on one box: with the product library and with the hipModule form (DABX_LIB + LD_LIBRARY_PATH); logs under gpurun_out/r06_suites/.
O=gpurun_out/r06_suites; mkdir -p $O
timeout 2400 python3 -m pytest tests -m gpu -q --maxfail=10 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -8 $O/pytest_gpu.log
DABX_LIB=$PWD/dabstar_amd/hipmodule/libdabx.so LD_LIBRARY_PATH=$PWD/dabstar_amd/hipmodule:$LD_LIBRARY_PATH timeout 2400 python3 -m pytest tests -m gpu -q --maxfail=10 > $O/pytest_gpu_hipmodule.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu_hipmodule.log; tail -8 $O/pytest_gpu_hipmodule.log
