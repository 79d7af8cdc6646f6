#!/bin/bash
cd /root/repo
for args in "8192 16384 0" "8192 16384 1" "4096 16384 0" "8192 4096 0" "8192 57344 0"; do timeout -k 10 120 python3 tools/lin_check.py $args 2>&1 | tail -4; done
