#!/bin/bash
# scratch for one-off gpurun experiments (see the exp_*.sh files for the kept ones)
echo nothing to run
