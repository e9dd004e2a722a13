for x in 0 8000 14000 20000 28000 40000; do CORA_SPMM_EXTRA_LDS=$x python tools/hvp_quick.py xlds_$x 2>&1 | tail -1; done
