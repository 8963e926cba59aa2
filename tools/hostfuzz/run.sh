#!/bin/bash
# Host-side sanitizer run of the parser / geometry / tap-table code (no GPU involved).
set -e
cd "$(dirname "$0")"
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=all -o /tmp/tts_parse_fuzz parse_fuzz.cpp ../../tiktoken_amd/csrc/filter.cpp ../../tiktoken_amd/csrc/vf_parse.cpp
/tmp/tts_parse_fuzz "${1:-200000}"
