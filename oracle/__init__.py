"""TEST INFRASTRUCTURE ONLY: CPU oracle for the MI355X hot path (see m355_oracle.c and
torch_ref.py headers).  Never imported by segmentation-pipeline_amd/."""
