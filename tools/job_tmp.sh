#!/bin/bash
O=gpurun_out
RTO_CD_BASIS=25 bash tools/contention_check.sh 150 | cut -c1-330
RTO_CD_BASIS=16 bash tools/contention_check.sh 100 | cut -c1-330 | tail -n 3
