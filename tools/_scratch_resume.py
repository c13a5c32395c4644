import os, sys, subprocess, tempfile, json
import numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT,"tests")); sys.path.insert(0, ROOT)
import test_data_pipeline as T
tmp=tempfile.mkdtemp()
rng=np.random.default_rng(3)
specs={k:T._write_coco(tmp,k,4,rng,sizes=[(96,192)]*4) for k in ("source","target")}
yaml=os.path.join(ROOT,"configs/da_faster_rcnn/e2e_da_faster_rcnn_R_50_C4_cityscapes_to_foggy_cityscapes.yaml")
def run(out,weight,mi):
    os.makedirs(out,exist_ok=True)
    cmd=[sys.executable,os.path.join(ROOT,"tools","train_net_da.py"),"--config-file",yaml,"--source",",".join(specs["source"]),"--target",",".join(specs["target"]),"SOLVER.MAX_ITER",str(mi),"SOLVER.CHECKPOINT_PERIOD","2","DATALOADER.NUM_WORKERS","0","INPUT.MIN_SIZE_TRAIN","(96,)","INPUT.MAX_SIZE_TRAIN","192","MODEL.OUTPUT_DIR",out,"MODEL.WEIGHT",weight]
    r=subprocess.run(cmd,capture_output=True,text=True)
    print("\n".join(l[:160] for l in r.stderr.splitlines() if "iter" in l or "Loading" in l or "Saving" in l or "WEIGHT" in l))
run(os.path.join(tmp,"a"),"",3)
print("----")
run(os.path.join(tmp,"b"),os.path.join(tmp,"a","model_0000002.pth"),5)
