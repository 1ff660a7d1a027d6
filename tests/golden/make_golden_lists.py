"""Golden file lists: the reference's dataloader/list_file.py (needs only `os`, loaded from its file) run on the miniature
tree of deep360_tree.py; paths stored relative to the tree root.  Run in the dev container:
    python tests/golden/make_golden_lists.py
"""
import importlib.util
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import deep360_tree  # noqa: E402


def rel(obj, root):
  if isinstance(obj, str):
    return os.path.relpath(obj, root)
  return [rel(o, root) for o in obj]


def main():
  spec = importlib.util.spec_from_file_location('ref_list_file', '/root/reference/dataloader/list_file.py')
  ref = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(ref)
  out = {}
  with tempfile.TemporaryDirectory() as root:
    dataset, exported, _ = deep360_tree.build(root)
    for soiled in (False, True):
      tag = 'soiled' if soiled else 'clean'
      out['disparity_train/' + tag] = rel(ref.list_deep360_disparity_train(dataset, soiled), root)
      out['disparity_test/' + tag] = rel(ref.list_deep360_disparity_test(dataset, soiled), root)
      out['fusion_train/' + tag] = rel(ref.list_deep360_fusion_train(exported, dataset, soiled), root)
      out['fusion_test/' + tag] = rel(ref.list_deep360_fusion_test(exported, dataset, soiled), root)
  with open(os.path.join(HERE, 'deep360_lists.json'), 'w') as f:
    json.dump(out, f, indent=0, sort_keys=True)
  print({k: len(json.dumps(v)) for k, v in out.items()})


if __name__ == '__main__':
  main()
