"""Test infrastructure: CPU restatement (torch fp32) of the reference's hot path. See packnet_oracle.py."""
