#!/bin/bash
MLX_LIB_PATH=$PWD/tools/libmlease_hip_sprof.so python tools/small_profile.py | tail -2
