#!/bin/bash
echo "--- product"; python scratch/dbg_deep2.py 2>&1 | grep -v Warn
echo "--- debug lib, no sync"; RQAE_HIP_LIBRARY=tools/_build/d3dbg/librqae_hip.so python scratch/dbg_deep2.py 2>&1 | grep -v Warn
echo "--- debug lib, sync around the products"; RQAE_DEEP_SYNC=1 RQAE_HIP_LIBRARY=tools/_build/d3dbg/librqae_hip.so python scratch/dbg_deep2.py 2>&1 | grep -v Warn
