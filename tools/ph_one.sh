#!/bin/bash
# Rebuild ONE source of the phase-timing library (make phases built the rest) for a few marked threads:
#   tools/ph_one.sh img_mid3 0 448 960   ->  libvar_ph_t<thread>.so
set -e
cd "$(dirname "$0")/../voicecontrolledrobot-var_amd/csrc"
src=$1; shift
OBJS=$(for f in $(grep '^SRCS' Makefile | cut -d= -f2); do echo /tmp/var_ph/${f%.hip}.o; done)
for T in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off -DVAR_PHASES -DVAR_PH_THREAD=$T -c $src.hip -o /tmp/var_ph/${src}_t$T.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvar_ph_t$T.so $(echo $OBJS | sed "s#/tmp/var_ph/$src.o#/tmp/var_ph/${src}_t$T.o#") -ldl
done
