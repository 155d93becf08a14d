#!/bin/bash
# tools/host_sanitize.sh -- the threaded host BVH builders (bvh_build.cpp) under ThreadSanitizer and under
# AddressSanitizer + UBSan, on the CPU (no GPU code is involved; GPU sanitizers are not available on this pool).
set -e
cd "$(dirname "$0")/../cudaraytracer_amd/csrc"
for san in thread address,undefined; do
  g++ -std=c++17 -O1 -g -fsanitize=$san -ffp-contract=off -I. -I../../include -o /tmp/crt_host_sanitize ../../tools/host_sanitize.cpp bvh_build.cpp -lpthread
  echo "== -fsanitize=$san"; /tmp/crt_host_sanitize
done
