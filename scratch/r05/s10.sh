#!/bin/bash
bash scratch/r05/pmc.sh g1_old g1 64 rowgroup_form=0 && bash scratch/r05/pmc.sh g1_t16 g1 64 rowgroup_form=1,grouptile_fields=16 && bash scratch/r05/pmc.sh g1_t32 g1 64 rowgroup_form=1,grouptile_fields=32
