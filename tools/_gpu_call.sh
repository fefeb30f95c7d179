#!/bin/bash
cd /root/repo
python tools/memset_async_probe.py 2>&1 | grep -v amdgpu.ids
