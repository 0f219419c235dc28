#!/bin/bash
# the dictionary gather on three dictionaries: in LDS, too large for LDS, long values
set -e -o pipefail
timeout -k 10 120 python bench_util/profile_extra.py dictionary
DICT_ENTRIES=100000 timeout -k 10 120 python bench_util/profile_extra.py dictionary
DICT_ENTRIES=300 DICT_MAX_LEN=120 timeout -k 10 120 python bench_util/profile_extra.py dictionary
DICT_ENTRIES=2000 DICT_MAX_LEN=12 timeout -k 10 120 python bench_util/profile_extra.py dictionary
