#!/bin/bash
# probe the GPU box ONCE for a Rust toolchain / crates that could build bench/rapier_ref.
echo "== which"; which cargo rustc rustup 2>&1
echo "== versions"; cargo --version 2>&1; rustc --version 2>&1
echo "== cargo homes"; ls -la ~/.cargo ~/.rustup /usr/local/cargo /opt/rust* 2>&1 | head -30
echo "== registry"; ls ~/.cargo/registry 2>&1 | head
echo "== find"; find / -xdev \( -name 'cargo' -o -name 'rustc' -o -name '*.crate' -o -name 'librapier3d*' -o -name 'rapier3d*' \) 2>/dev/null | grep -v '^/proc' | head -20
echo "== network"; timeout 8 curl -sI https://index.crates.io 2>&1 | head -3; echo "curl rc=$?"
timeout 5 getent hosts crates.io; echo "dns rc=$?"
echo "== host"; nproc; lscpu | grep -E 'Model name|Socket|Core|Thread' 
