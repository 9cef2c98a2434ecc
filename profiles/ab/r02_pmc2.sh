R=$(pwd)
sed -i '/^SQ_INSTS_LDS\|^GRBM_GUI\|^SQ_VALU_MFMA\|^TA_TA\|^SQ_WAVES_EQ/d' profiles/pmc_kernel2.sh
GSH_MC_PACKED_BODY=1 bash profiles/pmc_kernel2.sh mcorr_kernel mc_u256 python $R/profiles/ab/mcorr_ab.py | tail -12
GSH_LIB_PATH=$R/build/variants/lib_t128.so bash profiles/pmc_kernel2.sh mcorr_kernel mc_u128 python $R/profiles/ab/mcorr_ab.py | tail -12
