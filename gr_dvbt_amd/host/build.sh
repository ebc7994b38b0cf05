#!/bin/bash
# Compile-and-link check of the C++ host mirror against the in-tree libdvbt_hip.so
set -e
here="$(cd "$(dirname "$0")" && pwd)"
for ex in rx_flowgraph_example rx_stream_example rx_multi_example rx_blocks_bench; do
  g++ -std=c++17 -O2 -Wall -pthread -o "$here/$ex" "$here/$ex.cpp" -L"$here/../lib" -ldvbt_hip -Wl,-rpath,"$here/../lib" -Wl,-rpath,/opt/rocm/lib
  echo "built $here/$ex"
done
