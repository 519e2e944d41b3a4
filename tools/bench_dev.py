#!/usr/bin/env python3
"""bench.py on the DEVELOPMENT library with its switches set from the environment (the product library reads no environment):
    RDETR_LIB_PATH=relation_detr_amd/librelation_detr_amd_dev.so RDETR_DEV_RES_WAVES=8 python3 tools/bench_dev.py --no-cpu-baseline --no-extras"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from relation_detr_amd import _lib  # noqa: E402

lib = ctypes.CDLL(_lib.LIB_PATH)
for env, fn in (("RDETR_DEV_RES_MAX_TEAMS", "rdetr_dev_set_res_max_teams"), ("RDETR_DEV_RES_WAVES", "rdetr_dev_set_res_waves"),
                ("RDETR_DEV_RES_TILED", "rdetr_dev_set_res_tiled"), ("RDETR_DEV_HEAD_GROUP_LOG2", "rdetr_dev_set_msda_head_group_log2")):
    if os.environ.get(env):
        getattr(lib, fn)(int(os.environ[env]))
import bench  # noqa: E402

bench.main()
