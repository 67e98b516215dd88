#!/bin/bash
# Development aid (no GPU needed): the host-side builders under AddressSanitizer + UndefinedBehaviorSanitizer, and the threaded octree
# construction under ThreadSanitizer, on random triangle soups (tools/experiments/host_sanitize.cpp).
# usage: tools/sanitize_host.sh [cases] [seed]
set -e
cd "$(dirname "$0")/.."
src="tools/experiments/host_sanitize.cpp raytracer_amd/csrc/rt_host_build.cpp"
g++ -O1 -g -std=c++17 -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined -ffp-contract=off $src -o /tmp/host_asan -pthread
g++ -O1 -g -std=c++17 -fno-omit-frame-pointer -fsanitize=thread -ffp-contract=off $src -o /tmp/host_tsan -pthread
# (the harness's preloaded library is not a sanitizer runtime: the sanitizers' own must come first)
ASAN_OPTIONS=verify_asan_link_order=0:detect_leaks=1 /tmp/host_asan "${1:-40}" "${2:-1}"
TSAN_OPTIONS=halt_on_error=1 /tmp/host_tsan "${1:-40}" "${2:-1}"
