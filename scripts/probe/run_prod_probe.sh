#!/bin/bash
cd "$(dirname "$0")"
for v in 1; do for w in 1 4; do timeout -k 5 60 ./prod_probe $v $w 40 || echo "FAILED $v $w"; done; done
