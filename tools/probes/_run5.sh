bash tools/probes/ab.sh "DADET_STREAMK=1 DADET_STREAMK=0 DADET_STREAMK_SMALL=0" "img_only da fpn_dcn_da"
