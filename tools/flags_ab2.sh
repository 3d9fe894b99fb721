#!/bin/bash
# A/B of two switches under the bench arrangement
python tools/flag_ab.py emip_amd.model.EMIP_short.create_backbone KSPLIT False True
python tools/flag_ab.py emip_amd.lib.pvt_v2 SR_KSPLIT False True
