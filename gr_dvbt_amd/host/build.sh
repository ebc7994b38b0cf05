#!/bin/bash
# Compile-and-link check of the C++ host mirror against the in-tree libdvbt_hip.so
set -e
here="$(cd "$(dirname "$0")" && pwd)"
g++ -std=c++17 -O2 -Wall -o "$here/rx_flowgraph_example" "$here/rx_flowgraph_example.cpp" \
    -L"$here/../lib" -ldvbt_hip -Wl,-rpath,"$here/../lib" -Wl,-rpath,/opt/rocm/lib
echo "built $here/rx_flowgraph_example"
