"""utils.config — YAML + Jinja2 configuration (mirror of the reference module of the same name).

Same entry points and result as the reference: `get_render_config(path)` parses the YAML once to collect template
parameters, renders the same file as a Jinja2 template, parses the result and passes it through
`get_dump_config` (utils/config.py:12-93), which picks the output directory `<RESULT><model>__<n>_output/`,
coerces the float fields and creates the directory.  Deliberate differences:
  * the template that is rendered is the file at `path` (the reference always re-opens './config.yml', :18);
  * the `dqtl:` section is optional (the shipped reference config.yml lacks it and fails its own loader);
  * `yaml.safe_load` instead of `yaml.load(FullLoader)`;
  * earlier output directories without a result sheet are removed only when `delete: 1` (reference semantics, :55-71).
"""
import os
import shutil
from pathlib import Path

import yaml
from jinja2 import Template


def get_config(path):
    with open(path, encoding='utf-8') as f:
        return yaml.safe_load(f)


def _template_values(raw):
    """Values the YAML file refers to as {{name}} (reference: utils/config.py:20-35)."""
    city = raw['data_city']
    stage = raw.get('dqtl') or {}
    return dict(
        parameter1='value1', p2=Path(__file__).resolve().parent.parent, dc=city,
        num=len(raw['DATA_DICT'][city]['color']),                 # colours incl. the background class 0 (:25)
        tr=raw['train_rate'], ep=raw['epoch'], bs=raw['batchsize'],
        expo_result=raw['expo_result'], parameters=raw['parameters'], mn=raw['model_name'], FN=raw['FILE_NUM'],
        ne=stage.get('num_epochs', 0), ps=stage.get('pic_size', 0))


def get_render_config(path):
    raw = get_config(path)
    with open(path, 'r', encoding='utf-8') as f:
        rendered = Template(f.read()).render(**_template_values(raw))
    return get_dump_config(yaml.safe_load(rendered))


class _RunSlots:
    """Numbered result slots `<RESULT><model>__<n>_result.xlsx` / `<RESULT><model>__<n>_output/` (:44-77)."""

    def __init__(self, cfg):
        self.stem = cfg['RESULT'] + cfg['model_name'] + '__'

    def sheet(self, n):
        return '%s%d_result.xlsx' % (self.stem, n)

    def folder(self, n):
        return '%s%d_output/' % (self.stem, n)

    def taken(self, n):
        return os.path.exists(self.sheet(n)) or os.path.exists(self.folder(n))

    def first_free(self):
        n = 0
        while self.taken(n):
            n += 1
        return n


def get_dump_config(y):
    os.makedirs(y['RESULT'], exist_ok=True)
    slots = _RunSlots(y)
    if y['train']['index'] != 0:
        probe = slot = slots.first_free()
        if y.get('delete'):
            # an output folder without its result sheet is an aborted run: drop it and reuse the lowest such number
            for n in reversed(range(slot)):
                if os.path.isdir(slots.folder(n)) and not os.path.isfile(slots.sheet(n)):
                    shutil.rmtree(slots.folder(n))
                    slot = n
        y['FILE_NUM'] = slot
    else:
        probe = 0
        slot = y['FILE_NUM']
    y['RESULT_excel'] = slots.sheet(probe)                  # as the reference: the name probed last (:77)
    y['RESULT_output'] = slots.folder(slot)
    for key in ('lr', 'base_lr'):
        y['schedule'][key] = float(y['schedule'][key])
    y['Categories_Number'] = int(y['Categories_Number'])
    stage = y.get('dqtl') or {}
    for key in ('lr', 'tao', 'epsilon'):
        if key in stage:
            stage[key] = float(stage[key])
    y = yaml.safe_load(yaml.dump(y))
    if y['train']['save_best'] and not os.path.exists(y['RESULT_output']):
        os.makedirs(y['RESULT_output'])
    return y
