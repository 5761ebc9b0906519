timeout -k 10 400 python tools/other_configs.py 2>&1 | grep -E "configs|peak|Error|error"
